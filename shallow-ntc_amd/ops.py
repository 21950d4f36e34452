"""Operator layer: torch-ROCm tensors as buffers, libsntc_hip.so as the arithmetic.

Every function takes / returns dense NHWC float32 CUDA tensors, passes raw device pointers and the
current HIP stream through the C ABI (include/sntc.h), and allocates outputs with torch's caching
allocator.  No torch compute op is used on the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _capi as capi

ACTS = {None: capi.ACT_NONE, "none": capi.ACT_NONE, "relu": capi.ACT_RELU, "leaky_relu": capi.ACT_LEAKY_RELU,
        "lrelu": capi.ACT_LEAKY_RELU, "sigmoid": capi.ACT_SIGMOID}
KINDS = {"conv": capi.CONV2D, "convT": capi.CONV2D_TRANSPOSE, "sigdown": capi.SIGNAL_DOWN, "sigup": capi.SIGNAL_UP}


WGRAD_STREAM = None         # training: the side stream the weight / bias gradients of TConv layers are launched on (Trainer sets it)
FUSE_RESIDUAL_TAIL = True   # ResidualBlock (c = 192): 3x3 and 1x1 + skip in one launch (bit-identical; False: two launches)
FUSE_RESIDUAL_BLOCK = not os.environ.get("SNTC_NO_RB_FUSE")   # ResidualBlock (c = 192): head, 3x3 and tail + skip in ONE launch on an 8 x 32
                            # pixel tile with its halo patch in LDS (csrc/rb_fused.hip; bit-identical to the three launches)
FUSED_BLOCK_MIN_TILES = 256  # ... where the launch offers at least one 8 x 32 tile per CU (one workgroup per CU); below, the layers
CONCURRENT_BRANCHES = not os.environ.get("SNTC_NO_BRANCH_STREAMS")   # SimpleAttention: trunk and branch on two streams when the launches are small
CONCURRENT_BRANCH_MAX_TILES = 1024  # ... i.e. up to a few 8 x 32 pixel tiles per CU (two launches then share the rounds a single one leaves ragged)
MAX_INPUT_BYTES = 1 << 31   # sntc_conv_forward: inputs are addressed with 32-bit buffer offsets
PROFILE = None   # set to a list to record one entry per convolution launch (bench.py)
ROW_PACKED_FIRST_LAYER = not os.environ.get("SNTC_NO_ROWPACK")    # Cin = 3 analysis layers run as row-packed plans (False: the generic dword-gather path, for the A/B)
FORCE_TILE = 0   # tools/profile_layers.py --variant: every plan created afterwards is pinned to this tile variant
BF16X3_EXPERIMENT = False   # bench.py regions.decode_bf16x3 only: plans created while this is set use the split-precision
                            # contraction wherever it applies (Cin % 16 == 0, no prologue); never set by the product paths


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SIDE_POOL = {}
SIDE_POOL_MIN = 3


def side_streams(n, device=None):
    """The first ``n`` of the library's side streams on ``device`` (default: the current one): ONE pool per device, shared by
    everything that keeps launches in flight side by side -- the two-stream decode / encode of a set, ``Model.decode_set``, the
    look-ahead of ``Model.evaluate``, the blobs of ``decompress_many``, the attention block's branch (``companion_stream``).  A
    process has FOUR hardware queues per device by default (``GPU_MAX_HW_QUEUES``); the default stream holds one, the first three
    streams a process creates get one each, and every further stream SHARES a queue with an earlier one -- its kernels then wait
    for that stream's, and an event recorded behind them waits as well (``tools/microbench/hw_queues.py`` prints the map; with
    every component creating streams of its own, the pipelined ``decompress_many`` ran 6.35 ms in ``bench.py``'s process and
    5.37 ms in a process with three streams).  The pool's first three streams are created together, on first use; asking for more
    than three works and shares queues (raising ``GPU_MAX_HW_QUEUES`` instead costs the two-stream decode 5 %)."""
    if device is None:
        device = torch.cuda.current_device()
    idx = device.index if isinstance(device, torch.device) else int(device)
    if idx is None:
        idx = torch.cuda.current_device()
    pool = _SIDE_POOL.setdefault(idx, [])
    while len(pool) < max(int(n), SIDE_POOL_MIN):
        pool.append(torch.cuda.Stream(device=idx))
    return pool[:int(n)]


_COMPANIONS = {}


def companion_stream():
    """A second stream paired with the CURRENT one (one per current stream and device, created on first use -- AFTER the pool
    above, so that it never takes one of the pool's hardware queues: it shares one, which costs nothing where it is used, a lone
    image's attention branches with the pool idle): independent sub-graphs of a small launch run on it next to the caller's
    stream.  None while a HIP graph is being captured or a schedule is being measured (ops.autotune times launches with the
    device to itself)."""
    if AUTOTUNE or torch.cuda.is_current_stream_capturing():
        return None
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    st = _COMPANIONS.get(key)
    if st is None:
        side_streams(SIDE_POOL_MIN, cur.device)
        st = _COMPANIONS[key] = torch.cuda.Stream(device=cur.device)
    return st


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _check_nhwc(x, c=None):
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous()):
        raise ValueError("expected a contiguous float32 NHWC CUDA tensor, got "
                         f"{type(x).__name__} {getattr(x, 'dtype', None)} {tuple(getattr(x, 'shape', ()))}")
    if c is not None and x.shape[-1] != c:
        raise ValueError(f"expected {c} channels, got {x.shape[-1]}")


def _check_s3(x, c=None):
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 6 and x.is_contiguous()
            and tuple(x.shape[4:]) == (3, 16)):
        raise ValueError("expected a contiguous bfloat16 CUDA tensor in format S3 [n, h, w, C / 16, 3, 16] (ops.split3), got "
                         f"{type(x).__name__} {getattr(x, 'dtype', None)} {tuple(getattr(x, 'shape', ()))}")
    if c is not None and x.shape[3] * 16 != c:
        raise ValueError(f"expected {c} channels, got {x.shape[3] * 16}")


def split3(x):
    """fp32 NHWC -> format S3 (include/sntc.h): per pixel and 16-channel slab [hi | mid | lo] bfloat16, x = hi + mid + lo."""
    _check_nhwc(x)
    n, h, w, c = x.shape
    if c % 16:
        raise ValueError(f"format S3 needs a channel count divisible by 16, got {c}")
    out = torch.empty((n, h, w, c // 16, 3, 16), dtype=torch.bfloat16, device=x.device)
    capi.call("sntc_split3", _ptr(x), n * h * w, c, _ptr(out), _stream())
    return out


def dequant_split3(symbols, hyper, want_float=False):
    """y_hat = symbols + mu in format S3 (the decoder's dequantisation fused with the split); also fp32 if asked."""
    c = symbols.shape[-1]
    _check_nhwc(hyper, 2 * c)
    n, h, w, _ = symbols.shape
    out = torch.empty((n, h, w, c // 16, 3, 16), dtype=torch.bfloat16, device=symbols.device)
    y_hat = torch.empty(symbols.shape, dtype=torch.float32, device=symbols.device) if want_float else None
    capi.call("sntc_dequant_split3", _ptr(symbols), _ptr(hyper), n * h * w, c, _ptr(out), _ptr(y_hat), _stream())
    return (out, y_hat) if want_float else out


class PlanGroup:
    """Plans re-packed together by one launch (sntc_plan_group): ``entries`` = [(ConvPlan, weight tensor, bias tensor or None)].
    The tensors must stay where they are (views of the trainer's flat store); ``update()`` re-reads them."""

    def __init__(self, entries):
        capi.require_gpu()
        self._keep = list(entries)
        n = len(self._keep)
        plans = (C.c_void_p * n)(*[e[0]._h.value for e in self._keep])
        ws = (C.c_void_p * n)(*[e[1].data_ptr() for e in self._keep])
        bs = (C.c_void_p * n)(*[(e[2].data_ptr() if e[2] is not None else None) for e in self._keep])
        self._h = C.c_void_p()
        capi.call("sntc_plan_group_create", plans, ws, bs, n, _stream(), C.byref(self._h))

    def update(self):
        capi.call("sntc_plan_group_update", self._h, _stream())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                capi.load().sntc_plan_group_destroy(h)
            except Exception:
                pass
            self._h = None


class ResBlockPlan:
    """One whole ResidualBlock (reference common/elic.py:41-68) as one launch: sntc_resblock_plan (csrc/rb_fused.hip).
    ``w0`` [1,1,c,c/2], ``w1`` [3,3,c/2,c/2], ``w2`` [1,1,c/2,c] are the Keras kernels, ``b*`` the biases or None."""

    @staticmethod
    def supported(c):
        return bool(capi.load().sntc_resblock_supported(int(c)))

    def __init__(self, w0, b0, w1, b1, w2, b2, precision="fp32"):
        """``precision`` "bf16x3": the split-precision instantiation (csrc/rb_fused_bf3.hip) -- fp32-level accuracy, not
        bit-identical to "fp32"; fp32 tensors in and out either way."""
        capi.require_gpu()
        if precision not in ("fp32", "bf16x3"):
            raise ValueError(f"precision {precision!r}")
        self.precision = precision
        self.c = int(w0.shape[2])
        if tuple(w0.shape) != (1, 1, self.c, self.c // 2) or tuple(w1.shape) != (3, 3, self.c // 2, self.c // 2) \
                or tuple(w2.shape) != (1, 1, self.c // 2, self.c):
            raise ValueError(f"ResidualBlock kernels {tuple(w0.shape)}, {tuple(w1.shape)}, {tuple(w2.shape)} do not form a block")
        ts = [None if t is None else t.contiguous() for t in (w0, b0, w1, b1, w2, b2)]
        self._h = C.c_void_p()
        capi.call("sntc_resblock_plan_create", self.c, *[_ptr(t) for t in ts], int(precision == "bf16x3"), _stream(), C.byref(self._h))
        torch.cuda.current_stream().synchronize()   # packing reads the arrays; they may be freed after this

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                capi.load().sntc_resblock_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    def update(self, w0, b0, w1, b1, w2, b2):
        capi.call("sntc_resblock_plan_update", self._h, *[_ptr(t) for t in (w0, b0, w1, b1, w2, b2)], _stream())

    def set_workgroups(self, n):
        """Cap the persistent workgroups of this plan's launches (0: one per CU); tests: identical bits for any value."""
        capi.call("sntc_resblock_plan_set_workgroups", self._h, int(n))

    def flops(self, n, h, w):
        return int(capi.load().sntc_resblock_flops(self._h, n, h, w))

    @staticmethod
    def tiles(n, h, w):
        return n * (-(-h // 8)) * (-(-w // 32))

    def __call__(self, x):
        _check_nhwc(x, self.c)
        n, h, w, _ = x.shape
        if x.numel() * 4 >= MAX_INPUT_BYTES and n > 1:
            half = n // 2
            return torch.cat([self(x[:half]), self(x[half:])])
        y = torch.empty_like(x)
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        capi.call("sntc_resblock_forward", self._h, _ptr(x), n, h, w, _ptr(y), _stream())
        if prof is not None:
            e1.record()
            prof.append(dict(e0=e0, e1=e1, flops=self.flops(n, h, w), variant=0, nblocks=0, vec=True,
                             kind="resblock" if self.precision == "fp32" else "resblock3", k=3, s=1,
                             cin=self.c, cout=self.c, n=n, h=h, w=w))
        return y


FUSED_SYNTHESIS_MIN_ITEMS = 512   # ... where the launch offers about two items per CU (speed only: the same bits either way)
FUSED_SYNTHESIS_MIN_COLUMNS = 24  # ... and the layer has at least 24 output columns per phase ((1 + has_res) x hidden channels): with 12
                                  # (TwoLayerSynthesis(12, 3), the default of two_layer_syn2) a unit of 96 columns is eight phases whose shift
                                  # sets pad each other -- 86 against 108 TFLOP/s for the gather GEMM, whose [base] output the tail kernel
                                  # activates in its own stage 1 anyway: decode of 5 x 1200 x 1200 2.36 -> 2.27 ms without it (round 6)
FUSED_SYNTHESIS = not os.environ.get("SNTC_NO_SYN_FUSE")   # two-layer syntheses: first layer + activation + residual in ONE launch (csrc/syn_fused.hip;
                                                           # bit-identical; False: phase-grouped gather GEMM + the tail kernel's stage 1)


class SynPlan:
    """The first layer of a two-layer synthesis (reference common/transforms.py:298-361) as one launch: sntc_syn_plan
    (csrc/syn_fused.hip).  ``w1`` [k, k, (1 + has_res) ch, cin]: the Keras transposed kernel, the convolutional residual
    branch's concatenated behind the base convolution's on the output-channel axis; ``b1`` likewise or None; ``beta`` [ch],
    ``gamma`` [ch, ch] for the (I)GDN1 activations (``act_kind`` as ops.TAIL_ACTS)."""

    @staticmethod
    def supported(k, stride, cin, ch, has_res):
        return FUSED_SYNTHESIS and bool(capi.load().sntc_syn_supported(int(k), int(stride), int(cin), int(ch), int(bool(has_res))))

    def __init__(self, w1, b1, stride, ch, has_res, act_kind, beta=None, gamma=None):
        capi.require_gpu()
        self.k, self.stride, self.ch, self.has_res, self.act_kind = int(w1.shape[0]), int(stride), int(ch), bool(has_res), int(act_kind)
        self.cin = int(w1.shape[3])
        if tuple(w1.shape) != (self.k, self.k, ch * (2 if has_res else 1), self.cin):
            raise ValueError(f"synthesis kernel {tuple(w1.shape)} does not match ch = {ch}, has_res = {has_res}")
        ts = [None if t is None else t.contiguous() for t in (w1, b1, beta, gamma)]
        self._h = C.c_void_p()
        capi.call("sntc_syn_plan_create", self.k, self.stride, self.cin, self.ch, int(self.has_res), self.act_kind,
                  *[_ptr(t) for t in ts], _stream(), C.byref(self._h))
        torch.cuda.current_stream().synchronize()   # packing reads the arrays; they may be freed after this
        self._nunits = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                capi.load().sntc_syn_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    def update(self, w1, b1, beta=None, gamma=None):
        capi.call("sntc_syn_plan_update", self._h, *[_ptr(t) for t in (w1, b1, beta, gamma)], _stream())

    def set_workgroups(self, n):
        """Cap the persistent workgroups of this plan's launches (0: one per CU); tests: identical bits for any value."""
        capi.call("sntc_syn_plan_set_workgroups", self._h, int(n))

    def units(self):
        """[(shifts per channel slab, phases held, 32-row tile steps per slab, partial-step mask, [shift indices])] of the plan's
        units, in processing order."""
        buf = (C.c_int * (4 * 128))()
        n = int(capi.load().sntc_syn_plan_units(self._h, buf, 128))
        out = []
        for i in range(n):
            ns, word, lo, hi = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2] & 0xffffffff, buf[4 * i + 3] & 0xffffffff
            out.append((ns, word & 255, (word >> 8) & 255, (word >> 16) & 0xffff,
                        [((lo >> (4 * j)) if j < 8 else (hi >> (4 * (j - 8)))) & 15 for j in range(ns)]))
        return out

    def flops(self, latent_pixels):
        return int(capi.load().sntc_syn_flops(self._h, int(latent_pixels)))

    def fits(self, x):
        """Whether this call shape is inside the kernel's limits (latent rows of at most 127 pixels, one image < 2 GiB)."""
        n, h, w, c = x.shape
        return w <= 127 and h * w * max(c, self.stride * self.stride * self.ch) * 4 < (1 << 31)

    def items(self, xs):
        """Work items (256-pixel tiles x units) of a call on these batches: below FUSED_SYNTHESIS_MIN_ITEMS the launch cannot
        fill the device (one workgroup per CU, an item is 25 - 100 us) and the layers' small tiles are faster."""
        if self._nunits is None:
            self._nunits = len(self.units())
        return sum(int(x.shape[0]) * (-(-int(x.shape[1]) * int(x.shape[2]) // 256)) for x in xs) * self._nunits

    def __call__(self, xs):
        """``xs``: one y_hat tensor [n, h, w, cin] or a list of up to four of DIFFERENT image sizes (one launch for all of
        them) -> the hidden tensor(s) [n, stride h, stride w, ch]."""
        single = torch.is_tensor(xs)
        xs = [xs] if single else list(xs)
        if not 1 <= len(xs) <= 4:
            raise ValueError("1 .. 4 batches per call")
        outs = []
        arr = (capi.SynBatch * len(xs))()
        px = 0
        for i, x in enumerate(xs):
            _check_nhwc(x, self.cin)
            n, h, w, _ = x.shape
            y = torch.empty((n, h * self.stride, w * self.stride, self.ch), dtype=torch.float32, device=x.device)
            outs.append(y)
            arr[i].y_hat, arr[i].hidden, arr[i].n, arr[i].h, arr[i].w = x.data_ptr(), y.data_ptr(), n, h, w
            px += n * h * w
        dev = xs[0].device
        ws = torch.empty((64,), dtype=torch.int32, device=dev)     # the launch's work queue: private to this call
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        capi.call("sntc_syn_forward", self._h, arr, len(xs), _ptr(ws), 256, _stream())
        if prof is not None:
            e1.record()
            prof.append(dict(e0=e0, e1=e1, flops=self.flops(px), variant=0, nblocks=0, vec=True, kind="synthesis", k=self.k,
                             s=self.stride, cin=self.cin, cout=self.ch * (2 if self.has_res else 1), n=sum(x.shape[0] for x in xs),
                             h=xs[0].shape[1], w=xs[0].shape[2]))
        return outs[0] if single else outs


def to_device(a, device):
    """Host array -> float32 device tensor (weights, images)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


_PLAN_REGISTRY = []     # weak references to every ConvPlan in creation order: the plan's index is its identity across the ranks of
                        # a job (every rank builds the same models in the same order) -- export_tuning() / import_tuning()


class ConvPlan:
    """One packed convolution (sntc_conv_plan): Conv2D / Conv2DTranspose / SignalConv2D (+GDN pool)."""

    def __init__(self, kind, weight, bias, stride, act=None, prologue=capi.PRO_NONE, epilogue=capi.EPI_STORE,
                 kernel_io_swapped=False, bf16x3=False, rowpack=False):
        """``kernel_io_swapped``: ``weight`` has its two channel axes swapped with respect to the kind's own layout -- the
        input-gradient plan of a tfc.SignalConv2D layer (the adjoint kind) packs straight from the layer's kernel array."""
        capi.require_gpu()
        w = weight
        kh, kw = int(w.shape[0]), int(w.shape[1])
        if (kind == "convT") != bool(kernel_io_swapped):
            cout, cin = int(w.shape[2]), int(w.shape[3])
        else:
            cin, cout = int(w.shape[2]), int(w.shape[3])
        self.kind, self.cin, self.cout, self.stride, self.k = kind, cin, cout, int(stride), (kh, kw)
        self.epilogue = epilogue
        desc = capi.ConvDesc(kind=KINDS[kind], kh=kh, kw=kw, stride=int(stride), cin=cin, cout=cout,
                             act=ACTS[act], prologue=prologue, epilogue=epilogue)
        if BF16X3_EXPERIMENT and cin % 16 == 0 and prologue == capi.PRO_NONE:
            bf16x3 = True
        self.bf16x3 = bool(bf16x3)
        self.s3 = bf16x3 == "presplit"                  # the input arrives in format S3 (ops.split3): csrc/bf3_gemm.hip
        self.rowpack = bool(rowpack)
        desc.reserved[0] = 1 if kernel_io_swapped else 0
        desc.reserved[1] = 2 if self.s3 else (1 if bf16x3 else 0)       # split precision, opt-in, never a default
        desc.reserved[2] = 1 if rowpack else 0          # row-packed small-Cin layer on a caller-padded input (RowPackedConv)
        w = w.contiguous()
        b = None if bias is None else bias.contiguous()
        self._h = C.c_void_p()
        capi.call("sntc_conv_plan_create", C.byref(desc), _ptr(w), _ptr(b), _stream(), C.byref(self._h))
        torch.cuda.current_stream().synchronize()   # packing reads w/b; they may be freed after this
        self._tuned = {}
        import weakref
        _PLAN_REGISTRY.append(weakref.ref(self))
        if FORCE_TILE:
            self.set_tile(FORCE_TILE)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                capi.load().sntc_conv_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    def update(self, weight, bias=None):
        """Re-pack from new device weights (training step); the arrays are read asynchronously on the current stream."""
        capi.call("sntc_conv_plan_update", self._h, _ptr(weight), _ptr(bias), _stream())

    def set_tile(self, variant):
        """Force this plan's gather-GEMM tile variant (0 = heuristic); profiling / tests."""
        capi.call("sntc_conv_plan_set_tile", self._h, int(variant))

    def set_stream_k(self, enabled, dma=None, force=False, halo=True, colm=None):
        """``enabled`` False forces the static one-workgroup-per-tile schedule for this plan; ``dma`` True / False forces the
        direct-to-LDS / register stage path (None: library default); ``force``: stream-K even where the default picks one
        workgroup per tile because the tiles are short; ``halo`` False (pre-split plans): stage every tap's activation rows
        separately instead of one patch per channel slab.  Tests: identical bits every way."""
        flags = int(bool(enabled)) | (0 if dma is None else (4 | (2 if dma else 0))) | (8 if force else 0) | (0 if halo else 16)
        flags |= 0 if colm is None else (32 | (64 if colm else 0))   # stream-K unit order: column tile / row strip outermost (None: rule)
        capi.call("sntc_conv_plan_set_schedule", self._h, flags)

    def fusable_with(self, second):
        """True if ``second`` (a 1x1 plan) can run behind this plan inside one launch (sntc_conv_forward_fused)."""
        return FUSE_RESIDUAL_TAIL and bool(capi.load().sntc_conv_fusable(self._h, second._h))

    def fused(self, second, x, res=None, aux=None):
        """second(self(x), res, aux) in ONE launch (the tail of a c = 192 ResidualBlock): the intermediate never leaves the
        registers; bit-identical to the two calls."""
        _check_nhwc(x, self.cin)
        n, h, w, _ = x.shape
        ho, wo = self.out_hw(h, w)
        y = torch.empty((n, ho, wo, second.cout), dtype=torch.float32, device=x.device)
        for t in (res, aux):
            if t is not None and tuple(t.shape) != tuple(y.shape):
                raise ValueError(f"epilogue operand shape {tuple(t.shape)} != output shape {tuple(y.shape)}")
        if (x.numel() * 4 >= MAX_INPUT_BYTES or y.numel() >= (1 << 32)) and n > 1:
            half = n // 2
            cut = lambda t, lo, hi: None if t is None else t[lo:hi]
            return torch.cat([self.fused(second, x[:half], cut(res, 0, half), cut(aux, 0, half)),
                              self.fused(second, x[half:], cut(res, half, n), cut(aux, half, n))])
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ws_bytes = int(capi.load().sntc_conv_fused_workspace_bytes(self._h, n, h, w))
        ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=x.device) if ws_bytes else None
        capi.call("sntc_conv_forward_fused", self._h, second._h, _ptr(x), n, h, w, _ptr(y), _ptr(res), _ptr(aux), _ptr(ws),
                  ws_bytes, _stream())
        if prof is not None:
            e1.record()
            prof.append(dict(e0=e0, e1=e1, flops=self.flops(n, h, w) + second.flops(n, ho, wo), variant=3, nblocks=0,
                             vec=True, kind=self.kind + "+1x1", k=self.k[0], s=self.stride, cin=self.cin, cout=second.cout,
                             n=n, h=h, w=w))
        return y

    def out_hw(self, h, w):
        ho, wo = C.c_int(), C.c_int()
        capi.call("sntc_conv_out_shape", self._h, h, w, C.byref(ho), C.byref(wo))
        return ho.value, wo.value

    def flops(self, n, h, w):
        """Algorithmic 2 * MAC FLOPs of a call.  A plan whose input channels were zero-padded by its owner (the SGA adjoint of the
        two-layer synthesis pads 24 gradient channels to 32 for the vector loader) counts the REAL channels only."""
        f = int(capi.load().sntc_conv_flops(self._h, n, h, w))
        alg = getattr(self, "algorithmic_cin", None)
        return f if alg is None else f * int(alg) // self.cin

    def launch_info(self, n, h, w):
        v, nb = C.c_int(), C.c_int()
        capi.call("sntc_conv_launch_info", self._h, n, h, w, C.byref(v), C.byref(nb))
        return v.value, nb.value

    def tune(self, x, res=None, aux=None, reps=3):
        """Time this plan's (tile, schedule) candidates on ``x`` and keep the fastest for calls of this shape
        (sntc_conv_plan_tune; identical bits whichever runs).  Returns (variant, stream_k)."""
        n, h, w = (int(v) for v in x.shape[:3])
        ho, wo = self.out_hw(h, w)
        y = torch.empty((n, ho, wo, self.cout), dtype=torch.float32, device=x.device)
        ws_bytes = int(capi.load().sntc_conv_tune_workspace_bytes(self._h, n, h, w))
        ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=x.device) if ws_bytes else None
        v, sk = C.c_int(), C.c_int()
        capi.call("sntc_conv_plan_tune", self._h, _ptr(x), n, h, w, _ptr(y), _ptr(res), _ptr(aux), _ptr(ws), ws_bytes, int(reps),
                  C.byref(v), C.byref(sk), _stream())
        self._tuned[(n, h, w)] = (v.value, sk.value)
        return v.value, sk.value

    def clear_tuning(self):
        capi.call("sntc_conv_plan_clear_tuning", self._h)
        self._tuned = {}

    def candidates(self, n, h, w):
        """[(variant, stream_k)] this plan could run a call of this shape with (all give identical bits)."""
        v, s = (C.c_int * 32)(), (C.c_int * 32)()
        k = int(capi.load().sntc_conv_plan_candidates(self._h, int(n), int(h), int(w), v, s, 32))
        if k < 0:
            raise capi.SntcError(capi.ERR_BAD_SHAPE, capi.last_error())
        return [(v[i], s[i]) for i in range(k)]

    def set_choice(self, n, h, w, variant, stream_k):
        """Record (variant, stream_k) as the schedule of calls of this shape (sntc_conv_plan_set_choice)."""
        capi.call("sntc_conv_plan_set_choice", self._h, int(n), int(h), int(w), int(variant), int(bool(stream_k)))
        self._tuned[(int(n), int(h), int(w))] = (int(variant), int(bool(stream_k)))

    def __call__(self, x, res=None, aux=None, out=None):
        if self.s3:
            _check_s3(x, self.cin)
        else:
            _check_nhwc(x, self.cin)
        n, h, w = x.shape[:3]
        ho, wo = self.out_hw(h, w)
        y = out if out is not None else torch.empty((n, ho, wo, self.cout), dtype=torch.float32, device=x.device)
        for t in (res, aux):
            if t is not None and tuple(t.shape) != tuple(y.shape):
                raise ValueError(f"epilogue operand shape {tuple(t.shape)} != output shape {tuple(y.shape)}")
        if x.numel() * x.element_size() >= MAX_INPUT_BYTES and n > 1:
            # the kernel addresses its input with 32-bit buffer offsets: split the batch (images are independent)
            half = n // 2
            r0 = None if res is None else res[:half]
            r1 = None if res is None else res[half:]
            a0 = None if aux is None else aux[:half]
            a1 = None if aux is None else aux[half:]
            self.__call__(x[:half], r0, a0, out=y[:half])
            self.__call__(x[half:], r1, a1, out=y[half:])
            return y
        if AUTOTUNE and (n, h, w) not in self._tuned:
            self.tune(x, res, aux, reps=AUTOTUNE)
        if LAUNCH_LOG is not None:
            LAUNCH_LOG.append((self, int(n), int(h), int(w)))
        prof = PROFILE
        if prof is not None:     # bench.py: HIP events on the launch stream around this kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ws_bytes = int(capi.load().sntc_conv_workspace_bytes(self._h, n, h, w))
        ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=x.device) if ws_bytes else None
        capi.call("sntc_conv_forward", self._h, _ptr(x), n, h, w, _ptr(y), _ptr(res), _ptr(aux), _ptr(ws), ws_bytes, _stream())
        if prof is not None:
            e1.record()
            v, nb = self.launch_info(n, h, w)
            colm = C.c_int()
            capi.call("sntc_conv_launch_order", self._h, n, h, w, C.byref(colm))
            prof.append(dict(e0=e0, e1=e1, flops=self.flops(n, h, w), variant=v, nblocks=nb, vec=self.cin % 16 == 0 or self.rowpack,
                             kind=self.kind, k=self.k[0], s=self.stride, cin=self.cin, cout=self.cout, n=n, h=h, w=w, colm=bool(colm.value)))
        return y


def take_conv_status():
    """The device's sticky stream-K status word, exchanged with 0 (sntc_conv_status; synchronises the current stream): non-zero
    = a launch since the last call gave up waiting for a hand-off and its results are invalid.  ``check_conv_status`` is the
    raising form; ``tune_step`` uses this one to drop a candidate schedule that does not hold beside the step's other streams."""
    flags = C.c_int(0)
    capi.call("sntc_conv_status", C.byref(flags), _stream())
    return int(flags.value)


def check_conv_status():
    """Call where the host synchronises anyway: raises if a stream-K launch since the last check gave up waiting for a
    neighbour's hand-off (an oversubscribed device; include/sntc.h "Stream-K health").  The schedule is switched to the
    static one for the rest of the process, so the caller can simply run the step again."""
    if take_conv_status():
        _latch_stream_k_off()
        raise capi.SntcError(capi.ERR_HIP, "a stream-K hand-off timed out (the device is shared with other streams / processes): "
                                           "the results since the last check are invalid; stream-K is now off, run the step again")


FUSED_TAIL_MIN_ROWS = 24576      # ResidualBlock tails smaller than this run as two launches (common/_graph.py::ResidualBlock)
LAUNCH_LOG = None       # a list: every ConvPlan call appends (plan, n, h, w) -- tune_step() uses it to find a step's launches
AUTOTUNE = False       # inside ``with ops.autotune():`` every ConvPlan measures its launch schedule the first time it sees a shape


class autotune:
    """``with ops.autotune(): model.decode(...)`` -- every convolution plan that runs inside the block times its candidate
    (tile, schedule) pairs on the tensors it is called with, once per (n, h, w), and keeps the fastest for later calls of that
    shape (sntc_conv_plan_tune).  All candidates give identical bits; run the block on ONE stream with the device otherwise
    idle, so that the timings mean something."""

    def __init__(self, reps=3):
        self.reps = int(reps)

    def __enter__(self):
        global AUTOTUNE
        self._old = AUTOTUNE
        AUTOTUNE = self.reps
        return self

    def __exit__(self, *exc):
        global AUTOTUNE
        AUTOTUNE = self._old
        return False


def tune_step(step_fn, reps=12, min_gain=0.004, max_launches=16, log=None, burst=1, passes=1):
    """Choose the schedules of a whole step by the step's own clock: ``step_fn`` (which may use several streams and must join
    them back into the current stream) is run once to find its convolution launches; then, launch by launch (longest
    contraction first), every (tile, schedule) candidate is tried and kept if the median time of ``reps`` samples improves by
    more than ``min_gain`` (otherwise the launch keeps what it had: a measured choice or the cost model's).  For steps whose
    launches overlap on the device -- where a schedule measured with the device to itself (``autotune``) can be the wrong
    one.  ``burst`` steps are launched back to back between the two events of one sample (a loop of steps overlaps the tail of
    one with the head of the next; a lone step does not); ``passes`` > 1 walks the launches again while the previous walk
    changed something (a launch's best schedule depends on what its neighbours run).  All candidates give identical bits.
    Returns (ms per step before, after)."""
    global LAUNCH_LOG

    def clock():
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(burst):
                step_fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / burst)
        ts.sort()
        return ts[len(ts) // 2]

    LAUNCH_LOG = []
    try:
        step_fn()
        torch.cuda.synchronize()
        launches = list(dict.fromkeys(LAUNCH_LOG))               # unique (plan, n, h, w), first-seen order
    finally:
        LAUNCH_LOG = None
    launches.sort(key=lambda e: -e[0].flops(e[1], e[2], e[3]))
    for _ in range(3):
        step_fn()
    check_conv_status()                      # what ran before this call is the caller's: raise here, not on a candidate
    base = clock()
    if take_conv_status():
        _latch_stream_k_off()
        raise capi.SntcError(capi.ERR_HIP, "tune_step: a stream-K hand-off of the step timed out with the schedules it came with; "
                                           "stream-K is now off, run the step again")
    rollback = []                            # every launch back to what it came with (first walk's view)
    for walk in range(max(1, int(passes))):
        changed = False
        for plan, n, h, w in launches[:max_launches]:
            if plan.rowpack:
                continue
            start = plan.launch_info(n, h, w)
            had = plan._tuned.get((n, h, w))

            def restore(plan=plan, n=n, h=h, w=w, had=had):
                if had is not None:
                    plan.set_choice(n, h, w, *had)
                else:
                    plan._tuned.pop((n, h, w), None)
                    capi.call("sntc_conv_plan_clear_tuning", plan._h)    # back to the cost model for every shape of this plan ...
                    for (nn, hh, ww), (vv, ss) in list(plan._tuned.items()):
                        capi.call("sntc_conv_plan_set_choice", plan._h, nn, hh, ww, vv, ss)      # ... but the ones already chosen

            if walk == 0:
                rollback.append(restore)

            ref = clock()                        # what the launch runs now, measured next to its candidates (the clock drifts)
            if take_conv_status():
                ref = float("inf")               # ... and it does not hold here: any candidate that does wins
            chosen, t_chosen = None, ref
            for v, sk in plan.candidates(n, h, w):
                plan.set_choice(n, h, w, v, sk)
                if plan.launch_info(n, h, w) == start and (had is None or (v, sk) == tuple(had)):
                    continue                                     # the launch `ref` was measured with
                t = clock()
                if t < t_chosen * (1.0 - min_gain):
                    t = max(t, clock())                          # a winner has to win twice
                if take_conv_status():
                    continue                                     # a hand-off timed out beside the step's other streams: not a candidate here
                if t < t_chosen * (1.0 - min_gain):
                    chosen, t_chosen = (v, sk), t
            if chosen is not None:
                plan.set_choice(n, h, w, *chosen)
                changed = True
            else:
                restore()
            if log is not None:
                log.append(dict(layer=f"{plan.kind} k{plan.k[0]} s{plan.stride} {plan.cin}->{plan.cout}", shape=[n, h, w], start=list(start),
                                chosen=None if chosen is None else list(chosen), ref_ms=round(ref, 4), step_ms=round(t_chosen, 4), walk=walk))
        if not changed:
            break
    for _ in range(3):                       # what was settled on has to hold together
        step_fn()
    if take_conv_status():
        for undo in rollback:
            undo()
        for _ in range(3):
            step_fn()
        if take_conv_status():
            _latch_stream_k_off()
            raise capi.SntcError(capi.ERR_HIP, "tune_step: stream-K hand-offs of the step time out; stream-K is now off, run the step again")
        if log is not None:
            log.append(dict(rolled_back=True))
        return base, base
    after = clock()                          # the settled step, measured again (not the last launch's last sample)
    if take_conv_status() or not (after == after and after != float("inf")):
        after = base
    return base, after


def export_tuning():
    """Every measured launch schedule of this process as plain data: [(plan index in creation order, kind, cin, cout, n, h, w,
    variant, stream_k)].  One rank measures (``with ops.autotune()``), the others ``import_tuning`` what it found, so that all
    ranks of a job run IDENTICAL launches (same bits either way -- every candidate computes the same chains -- but a rank
    that picked another schedule would be a skewed scaling point, and N ranks need not spend N x the tuning time)."""
    out = []
    for idx, ref in enumerate(_PLAN_REGISTRY):
        p = ref()
        if p is None:
            continue
        for (n, h, w), (v, sk) in sorted(p._tuned.items()):
            out.append((idx, p.kind, p.cin, p.cout, int(n), int(h), int(w), int(v), int(sk)))
    return out


def import_tuning(entries, strict=True):
    """Apply another rank's ``export_tuning()`` (or a file of an earlier run: bench.py --tuning-file); a plan whose (kind, cin,
    cout) does not match the entry's is a job whose ranks built different models: refused loudly -- unless ``strict`` is False
    (choices persisted by an earlier PROCESS: entries of plans that do not exist yet, or no longer match, are skipped; every
    candidate computes the same bits, so a stale entry can only cost speed).  Returns the number of entries applied."""
    applied = 0
    for idx, kind, cin, cout, n, h, w, v, sk in entries:
        if idx >= len(_PLAN_REGISTRY):
            if not strict:
                continue
            raise capi.SntcError(capi.ERR_BAD_SHAPE, f"import_tuning: plan {idx} does not exist here ({len(_PLAN_REGISTRY)} plans "
                                 "were built): the ranks did not build the same plans in the same order")
        p = _PLAN_REGISTRY[idx]()
        if p is None:
            continue        # collected here already (the cyclic collector runs at different times on different ranks): nothing to tune
        if (p.kind, p.cin, p.cout) != (kind, cin, cout):
            if not strict:
                continue
            raise capi.SntcError(capi.ERR_BAD_SHAPE, f"import_tuning: plan {idx} here is {(p.kind, p.cin, p.cout)}, the entry was "
                                 f"measured on {(kind, cin, cout)}: the ranks did not build the same plans in the same order")
        try:
            p.set_choice(n, h, w, v, sk)
            applied += 1
        except capi.SntcError:
            if strict:
                raise
    return applied


STREAM_K_TIMED_OUT = False      # latched by check_conv_status / tune_step when a hand-off timed out: stream-K stays off for the process


def _latch_stream_k_off():
    """A stream-K hand-off timed out: static schedules for the rest of the process ("run the step again"); nothing that merely
    restores an earlier setting (static_schedules.__exit__) may re-arm it."""
    global STREAM_K_TIMED_OUT
    STREAM_K_TIMED_OUT = True
    capi.call("sntc_conv_set_stream_k", 0)


def set_stream_k(enabled, force=False):
    """Process-wide default of the persistent stream-K schedule (bit-identical results either way).  After a hand-off has timed
    out in this process the switch stays off unless ``force`` (a caller that knows the device is its own again) is given."""
    global STREAM_K_TIMED_OUT
    if enabled and STREAM_K_TIMED_OUT and not force:
        return
    if enabled and force:
        STREAM_K_TIMED_OUT = False
    capi.call("sntc_conv_set_stream_k", int(bool(enabled)))


def stream_k_enabled():
    """The library's own switch (sntc_conv_get_stream_k), whoever set it last."""
    return bool(capi.load().sntc_conv_get_stream_k())


class static_schedules:
    """``with ops.static_schedules(): ...`` -- the convolutions launched inside take the one-workgroup-per-tile / split-K schedules
    instead of persistent stream-K workers (same bits): for launches that run BESIDE kernels which hold CUs for long -- the lone
    waves of an entropy decode with their tables in LDS -- where stream-K's workers would not all be resident.  The switch is the
    process-wide one (a host-side decision per launch); it is restored on exit -- unless a hand-off timed out inside the block
    (check_conv_status / tune_step latch stream-K off for the process: set_stream_k(True) is then a no-op)."""

    def __init__(self, active=True):
        self.active = bool(active)

    def __enter__(self):
        self._was = stream_k_enabled()
        if self.active and self._was:
            set_stream_k(False)
        return self

    def __exit__(self, *exc):
        if self.active and self._was:
            set_stream_k(True)
        return False


def gdn_small(x, beta, gamma, inverse=False, alpha=1, epsilon=1.0):
    _check_nhwc(x)
    c = x.shape[-1]
    y = torch.empty_like(x)
    capi.call("sntc_gdn_small", _ptr(x), x.numel() // c, c, _ptr(beta), _ptr(gamma), int(inverse), int(alpha),
              int(epsilon == 0.5), _ptr(y), _stream())
    return y


GDN_SMALL_CHANNELS = (4, 8, 12, 16, 24, 32, 48)
TAIL_CHANNELS = (12, 24, 48)
TAIL_ACTS = {None: 0, "none": 0, "igdn": 1, "igdn1": 1, "gdn": 2, "gdn1": 2, "relu": 3, "leaky_relu": 4, "lrelu": 4}


def _tail_profile(n, hh, wh, ch, cout, k2, s2):
    """bench.py's launch record of the tail kernel: the output layer's algorithmic FLOPs (Conv2DTranspose k2 x k2 / s2,
    ch -> cout: 2 * MACs of its input pixels, as sntc_conv_flops counts a transposed layer) between two events."""
    prof = PROFILE
    if prof is None:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    return prof, dict(e0=e0, e1=e1, flops=2 * n * hh * wh * k2 * k2 * ch * cout, variant=0, nblocks=0, vec=True, kind="tail",
                      k=k2, s=s2, cin=ch, cout=cout, n=n, h=hh, w=wh)


def _tail_profile_done(rec):
    if rec is not None:
        rec[1]["e1"].record()
        rec[0].append(rec[1])


def two_layer_tail(t, ch, has_res, act_kind, beta, gamma, w2, b2, k2=5, s2=2):
    _check_nhwc(t, ch * (2 if has_res else 1))
    n, hh, wh, _ = t.shape
    cout = int(w2.shape[2])
    y = torch.empty((n, hh * s2, wh * s2, cout), dtype=torch.float32, device=t.device)
    rec = _tail_profile(n, hh, wh, ch, cout, k2, s2)
    capi.call("sntc_two_layer_tail", _ptr(t), n, hh, wh, ch, int(has_res), act_kind, _ptr(beta), _ptr(gamma),
              _ptr(w2), _ptr(b2), k2, s2, cout, _ptr(y), _stream())
    _tail_profile_done(rec)
    return y


def two_layer_tail_pixels(t, ch, has_res, act_kind, beta, gamma, w2, b2, h, w, reference=None, k2=5, s2=2):
    """The same tail with crop + floats_to_pixels + quantize_image (+ integer SSE against ``reference``) fused into the
    launch: -> (uint8 [n, h, w, 3], int64 sse[n] or None).  No float reconstruction is written."""
    _check_nhwc(t, ch * (2 if has_res else 1))
    n, hh, wh, _ = t.shape
    cout = int(w2.shape[2])
    px = torch.empty((n, h, w, cout), dtype=torch.uint8, device=t.device)
    sse = None
    if reference is not None:
        _check_nhwc(reference, cout)
        if tuple(reference.shape) != (n, h, w, cout):
            raise ValueError(f"reference {tuple(reference.shape)} does not match the decoded size {(n, h, w, cout)}")
        sse = torch.empty((n,), dtype=torch.int64, device=t.device)
    rec = _tail_profile(n, hh, wh, ch, cout, k2, s2)
    capi.call("sntc_two_layer_tail_pixels", _ptr(t), n, hh, wh, ch, int(has_res), act_kind, _ptr(beta), _ptr(gamma), _ptr(w2),
              _ptr(b2), k2, s2, cout, h, w, _ptr(reference), _ptr(px), _ptr(sse), _stream())
    _tail_profile_done(rec)
    return px, sse


def pad_zero(x, top, left, bottom, right):
    """Zero padding on the two spatial axes (the explicit form of a convolution's SAME padding)."""
    _check_nhwc(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h + top + bottom, w + left + right, c), dtype=torch.float32, device=x.device)
    capi.call("sntc_pad_zero", _ptr(x), n, h, w, c, top, left, h + top + bottom, w + left + right, _ptr(y), _stream())
    return y


class RowPackedConv:
    """Keras Conv2D(k, s, SAME) with kw * Cin <= 16 -- the RGB first layer of every analysis transform (common/elic.py:147,
    common/transforms.py:183) -- as a row-packed plan: the image is zero-padded once (SAME), then each kernel row is one
    16-deep K stage read with the vector loader (include/sntc.h, sntc_conv_desc.reserved[2]).  Same products as the generic
    path, in another order of summation (k = row-major inside a kernel row either way; the zero slots add exact zeros)."""

    def __init__(self, weight, bias, stride, act=None):
        self.k, self.stride = int(weight.shape[0]), int(stride)
        self.plan = ConvPlan("conv", weight, bias, stride, act, rowpack=True)
        self.cin, self.cout = self.plan.cin, self.plan.cout

    def _pads(self, size):
        out = -(-size // self.stride)
        total = max((out - 1) * self.stride + self.k - size, 0)
        return total // 2, total - total // 2

    def out_hw(self, h, w):
        return -(-h // self.stride), -(-w // self.stride)

    def flops(self, n, h, w):
        ho, wo = self.out_hw(h, w)
        return 2 * n * ho * wo * self.k * self.k * self.cin * self.cout

    def __call__(self, x, res=None, aux=None):
        _check_nhwc(x, self.cin)
        (pt, pb), (pl, pr) = self._pads(x.shape[1]), self._pads(x.shape[2])
        return self.plan(pad_zero(x, pt, pl, pb, pr), res, aux)


RGB_FIRST_LAYER = not os.environ.get("SNTC_NO_RGBCONV")     # the RGB first layer on its own kernel (csrc/rgb_conv.hip); False: the row-packed plan (same bits)


class RgbConvPlan:
    """The RGB first layer of the analysis transforms (reference common/elic.py:147, common/transforms.py:183: Keras Conv2D(k, 2,
    SAME) on 3 channels) as one launch of its own kernel: sntc_rgbconv_plan (csrc/rgb_conv.hip) -- weights resident in LDS, the
    input patch double-buffered, no padded copy of the image; bit-identical to ``RowPackedConv`` (the same fma chains)."""

    @staticmethod
    def supported(k, stride, cin, cout, act=None, kind="conv"):
        return act in ACTS and kind in ("conv", "sigdown") and bool(
            capi.load().sntc_rgbconv_supported(KINDS[kind], int(k), int(stride), int(cin), int(cout), ACTS[act]))

    def __init__(self, weight, bias, stride, act=None, kind="conv"):
        """``kind`` "sigdown": tfc.SignalConv2D(corr=True, strides_down) -- MBT2018Analysis' first layer (reference
        common/transforms.py:152-155): the same kernel with the centred padding origin."""
        capi.require_gpu()
        self.kind = kind
        self.k, self.stride = int(weight.shape[0]), int(stride)
        self.cin, self.cout = int(weight.shape[2]), int(weight.shape[3])
        if weight.shape[0] != weight.shape[1]:
            raise ValueError(f"first-layer kernel {tuple(weight.shape)} is not square")
        w = weight.contiguous()
        b = None if bias is None else bias.contiguous()
        self._h = C.c_void_p()
        capi.call("sntc_rgbconv_plan_create", KINDS[kind], self.k, self.stride, self.cin, self.cout, _ptr(w), _ptr(b), ACTS[act], _stream(), C.byref(self._h))
        torch.cuda.current_stream().synchronize()   # packing reads the arrays; they may be freed after this

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                capi.load().sntc_rgbconv_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    def update(self, weight, bias=None):
        capi.call("sntc_rgbconv_plan_update", self._h, _ptr(weight), _ptr(bias), _stream())

    def set_workgroups(self, n):
        """Cap the persistent workgroups of this plan's launches (0: one per CU); tests: identical bits for any value."""
        capi.call("sntc_rgbconv_plan_set_workgroups", self._h, int(n))

    def out_hw(self, h, w):
        return -(-h // self.stride), -(-w // self.stride)

    def flops(self, n, h, w):
        return int(capi.load().sntc_rgbconv_flops(self._h, n, h, w))

    def __call__(self, x, res=None, aux=None):
        if res is not None or aux is not None:
            raise ValueError("the first-layer kernel has no epilogue operands")
        _check_nhwc(x, self.cin)
        n, h, w, _ = x.shape
        ho, wo = self.out_hw(h, w)
        if n > 1 and max(x.numel(), n * ho * wo * self.cout) * 4 >= MAX_INPUT_BYTES:
            half = n // 2
            return torch.cat([self(x[:half]), self(x[half:])])
        y = torch.empty((n, ho, wo, self.cout), dtype=torch.float32, device=x.device)
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        capi.call("sntc_rgbconv_forward", self._h, _ptr(x), n, h, w, _ptr(y), _stream())
        if prof is not None:
            e1.record()
            prof.append(dict(e0=e0, e1=e1, flops=self.flops(n, h, w), variant=0, nblocks=0, vec=True, kind="rgbconv", k=self.k, s=self.stride,
                             cin=self.cin, cout=self.cout, n=n, h=h, w=w))
        return y


SMALL_OUTPUT_LAYER = not os.environ.get("SNTC_NO_UPSMALL")   # the 5x5/2 transposed convolution to 3 channels on its own kernel (csrc/up_small.hip); False: the gather GEMM


class UpSmallPlan:
    """The last layer of the multi-layer syntheses (reference common/transforms.py:172-175 MBT2018Synthesis, :195-206
    CNNSynthesis: 5 x 5 / 2; :131-134 BLS2017Synthesis: 9 x 9 / 4): a transposed convolution to the 3 image channels as one launch of a vector-ALU kernel --
    sntc_upsmall_plan (csrc/up_small.hip); ``kind`` "convT" (Keras kernel [5, 5, 3, cin]) or "sigup" (tfc kernel [5, 5, cin, 3])."""

    @staticmethod
    def supported(kind, k, stride, cin, cout):
        return kind in ("convT", "sigup") and bool(capi.load().sntc_upsmall_supported(KINDS[kind], int(k), int(stride), int(cin), int(cout)))

    def __init__(self, kind, weight, bias, stride):
        capi.require_gpu()
        self.kind, self.stride = kind, int(stride)
        self.k = int(weight.shape[0])
        self.cout, self.cin = (int(weight.shape[2]), int(weight.shape[3])) if kind == "convT" else (int(weight.shape[3]), int(weight.shape[2]))
        w = weight.contiguous()
        b = None if bias is None else bias.contiguous()
        self._h = C.c_void_p()
        capi.call("sntc_upsmall_plan_create", KINDS[kind], self.k, self.stride, self.cin, self.cout, _ptr(w), _ptr(b), _stream(), C.byref(self._h))
        torch.cuda.current_stream().synchronize()   # packing reads the arrays; they may be freed after this

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                capi.load().sntc_upsmall_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    def update(self, weight, bias=None):
        capi.call("sntc_upsmall_plan_update", self._h, _ptr(weight), _ptr(bias), _stream())

    def out_hw(self, h, w):
        return h * self.stride, w * self.stride

    def flops(self, n, h, w):
        return int(capi.load().sntc_upsmall_flops(self._h, n, h, w))

    def __call__(self, x, res=None, aux=None):
        if res is not None or aux is not None:
            raise ValueError("the small-output transposed convolution has no epilogue operands")
        _check_nhwc(x, self.cin)
        n, h, w, _ = x.shape
        y = torch.empty((n, self.stride * h, self.stride * w, self.cout), dtype=torch.float32, device=x.device)
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        capi.call("sntc_upsmall_forward", self._h, _ptr(x), n, h, w, _ptr(y), _stream())
        if prof is not None:
            e1.record()
            prof.append(dict(e0=e0, e1=e1, flops=self.flops(n, h, w), variant=0, nblocks=0, vec=True, kind="upsmall", k=self.k, s=self.stride,
                             cin=self.cin, cout=self.cout, n=n, h=h, w=w))
        return y


def pad_reflect(x, hp, wp):
    _check_nhwc(x)
    n, h, w, c = x.shape
    if (hp, wp) == (h, w):
        return x
    y = torch.empty((n, hp, wp, c), dtype=torch.float32, device=x.device)
    capi.call("sntc_pad_reflect", _ptr(x), n, h, w, c, hp, wp, _ptr(y), _stream())
    return y


def concat_channels(a, b=None, extra=16):
    """[a | b] along the channel axis; with ``b`` None: a channel of ones and ``extra`` - 1 zero channels (sntc_concat_channels)."""
    _check_nhwc(a)
    n, h, w, ca = a.shape
    if b is not None:
        _check_nhwc(b)
        if tuple(b.shape[:3]) != (n, h, w):
            raise ValueError(f"concat_channels: {tuple(a.shape)} vs {tuple(b.shape)}")
    cb = int(b.shape[-1]) if b is not None else int(extra)
    y = torch.empty((n, h, w, ca + cb), dtype=torch.float32, device=a.device)
    capi.call("sntc_concat_channels", _ptr(a), ca, _ptr(b), cb, n * h * w, _ptr(y), _stream())
    return y


def depth_to_space(x, block=2):
    """tf.nn.depth_to_space in NHWC (sntc_depth_to_space)."""
    _check_nhwc(x)
    n, h, w, c = x.shape
    if c % (block * block):
        raise ValueError(f"depth_to_space: {c} channels are not a multiple of {block * block}")
    y = torch.empty((n, h * block, w * block, c // (block * block)), dtype=torch.float32, device=x.device)
    capi.call("sntc_depth_to_space", _ptr(x), n, h, w, c, int(block), _ptr(y), _stream())
    return y


def crop(x, h, w):
    _check_nhwc(x)
    n, hp, wp, c = x.shape
    if (hp, wp) == (h, w):
        return x
    y = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    capi.call("sntc_crop", _ptr(x), n, hp, wp, c, h, w, _ptr(y), _stream())
    return y


def to_pixels(x_hat, h, w):
    """Decoder's last step: crop to h x w, (v+.5)*255, round-half-even, saturate -> uint8 [n,h,w,c]."""
    _check_nhwc(x_hat)
    n, hs, ws, c = x_hat.shape
    px = torch.empty((n, h, w, c), dtype=torch.uint8, device=x_hat.device)
    capi.call("sntc_pixels_sse", _ptr(None), _ptr(x_hat), n, h, w, c, hs, ws, _ptr(px), _ptr(None), _stream())
    return px


def pixels_sse(x, x_hat, want_pixels=False):
    """uint8 quantisation of both images + per-image integer SSE.  x_hat may be spatially larger
    (the crop of unpad_images is fused)."""
    _check_nhwc(x)
    _check_nhwc(x_hat, x.shape[-1])
    n, h, w, c = x.shape
    hs, ws = x_hat.shape[1], x_hat.shape[2]
    sse = torch.empty((n,), dtype=torch.int64, device=x.device)
    px = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device) if want_pixels else None
    capi.call("sntc_pixels_sse", _ptr(x), _ptr(x_hat), n, h, w, c, hs, ws, _ptr(px), _ptr(sse), _stream())
    return sse, px


def float_sse(x, x_hat):
    _check_nhwc(x)
    _check_nhwc(x_hat, x.shape[-1])
    n, h, w, c = x.shape
    sse = torch.empty((n,), dtype=torch.float64, device=x.device)
    capi.call("sntc_float_sse", _ptr(x), _ptr(x_hat), n, h, w, c, x_hat.shape[1], x_hat.shape[2], _ptr(sse), _stream())
    return sse


class DeepFactorizedPrior:
    """Device copy of tfc.NoisyDeepFactorized parameters (sntc_prior)."""

    def __init__(self, matrices, biases, factors):
        capi.require_gpu()
        self.channels = int(matrices[0].shape[0])
        widths = [int(matrices[0].shape[2])] + [int(m.shape[1]) for m in matrices]
        nl = len(matrices)

        def flat(arrs):
            if not arrs:
                return None
            a = np.concatenate([np.ascontiguousarray(v, dtype=np.float32).ravel() for v in arrs])
            return a.ctypes.data_as(C.POINTER(C.c_float)), a

        pm, _km = flat(matrices)
        pb, _kb = flat(biases)
        pf_keep = flat(factors)
        warr = (C.c_int * (nl + 1))(*widths)
        self._h = C.c_void_p()
        capi.call("sntc_prior_create", self.channels, nl, warr, pm, pb, pf_keep[0] if pf_keep else None,
                  _stream(), C.byref(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                capi.load().sntc_prior_destroy(h)
            except Exception:
                pass
            self._h = None

    def __call__(self, z, values_only=False):
        """-> (z_hat, bits[n] float64).  values_only: evaluate bits at z itself (explicit sample)."""
        _check_nhwc(z, self.channels)
        n = z.shape[0]
        hw = z.shape[1] * z.shape[2]
        bits = torch.empty((n,), dtype=torch.float64, device=z.device)
        z_hat = None if values_only else torch.empty_like(z)
        capi.call("sntc_entropy_factorized", self._h, _ptr(z), n, hw, _ptr(z_hat), _ptr(bits), int(values_only), _stream())
        return (z if values_only else z_hat), bits


def entropy_scale_normal(y, hyper, want_symbols=False, values_only=False):
    """-> (y_hat, bits[n] float64, symbols int32 or None)."""
    _check_nhwc(y)
    c = y.shape[-1]
    _check_nhwc(hyper, 2 * c)
    if tuple(hyper.shape[:3]) != tuple(y.shape[:3]):
        raise ValueError(f"hyper-synthesis output {tuple(hyper.shape)} does not match latents {tuple(y.shape)}")
    n = y.shape[0]
    hw = y.shape[1] * y.shape[2]
    bits = torch.empty((n,), dtype=torch.float64, device=y.device)
    y_hat = None if values_only else torch.empty_like(y)
    sym = torch.empty(y.shape, dtype=torch.int32, device=y.device) if (want_symbols and not values_only) else None
    capi.call("sntc_entropy_scale_normal", _ptr(y), _ptr(hyper), n, hw, c, _ptr(y_hat), _ptr(sym), _ptr(bits),
              int(values_only), _stream())
    return (y if values_only else y_hat), bits, sym


def dequant_scale_normal(symbols, hyper):
    c = symbols.shape[-1]
    _check_nhwc(hyper, 2 * c)
    n = symbols.shape[0]
    hw = symbols.shape[1] * symbols.shape[2]
    y_hat = torch.empty(symbols.shape, dtype=torch.float32, device=symbols.device)
    capi.call("sntc_dequant_scale_normal", _ptr(symbols), _ptr(hyper), n, hw, c, _ptr(y_hat), _stream())
    return y_hat


# ------------------------------------------------------------------------------------------
# SGA iterative inference (include/sntc.h "SGA" section)
# ------------------------------------------------------------------------------------------
def sga_factorized_fwd(prior, z_loc, tau, noise=None, seed=0, step=0):
    """-> (z_tilde, sprime, dbits_dz, bits[n])."""
    _check_nhwc(z_loc, prior.channels)
    n, hw = z_loc.shape[0], z_loc.shape[1] * z_loc.shape[2]
    zt, sp, db = torch.empty_like(z_loc), torch.empty_like(z_loc), torch.empty_like(z_loc)
    bits = torch.empty((n,), dtype=torch.float64, device=z_loc.device)
    capi.call("sntc_sga_factorized_fwd", prior._h, _ptr(z_loc), n, hw, float(tau), _ptr(noise), int(seed), int(step),
              _ptr(zt), _ptr(sp), _ptr(db), _ptr(bits), _stream())
    return zt, sp, db, bits


def sga_normal_fwd(y_loc, hyper, tau, noise=None, seed=0, step=0):
    """-> (y_tilde, sprime, dbits_dv, dbits_draw, bits[n])."""
    _check_nhwc(y_loc)
    c = y_loc.shape[-1]
    _check_nhwc(hyper, 2 * c)
    n, hw = y_loc.shape[0], y_loc.shape[1] * y_loc.shape[2]
    yt, sp, dv, dr = (torch.empty_like(y_loc) for _ in range(4))
    bits = torch.empty((n,), dtype=torch.float64, device=y_loc.device)
    capi.call("sntc_sga_normal_fwd", _ptr(y_loc), _ptr(hyper), n, hw, c, float(tau), _ptr(noise), int(seed), int(step),
              _ptr(yt), _ptr(sp), _ptr(dv), _ptr(dr), _ptr(bits), _stream())
    return yt, sp, dv, dr, bits


def sga_normal_bwd(g_ytilde, sprime, dbits_dv, dbits_draw, weight):
    """-> (g_yloc, g_hyper[.., 2C])."""
    _check_nhwc(g_ytilde)
    n, h, w, c = g_ytilde.shape
    g_y = torch.empty_like(g_ytilde)
    g_h = torch.empty((n, h, w, 2 * c), dtype=torch.float32, device=g_ytilde.device)
    capi.call("sntc_sga_normal_bwd", _ptr(g_ytilde), _ptr(sprime), _ptr(dbits_dv), _ptr(dbits_draw), float(weight),
              n * h * w, c, _ptr(g_y), _ptr(g_h), _stream())
    return g_y, g_h


def sga_chain(g, dbits, sprime, weight):
    out = torch.empty_like(g)
    capi.call("sntc_sga_chain", _ptr(g), _ptr(dbits), _ptr(sprime), float(weight), g.numel(), _ptr(out), _stream())
    return out


UQ_MODES = {"round": 0, "unoise": 1, "sga": 2, "soft_round": 3}


def uq_sample(loc, offset=None, mode="round", param=0.0, noise=None, seed=0, step=0):
    """UQLatentRV.sample / .quantize (reference common/latent_rvs_lib.py:77-116) on an NHWC tensor.  ``offset``: None, a
    tensor shaped like ``loc``, the mean half ``hyper[..., :c]`` of a hyper-synthesis output (a strided view is read in
    place), or a per-channel [c] tensor."""
    _check_nhwc(loc)
    c = loc.shape[-1]
    npix = loc.numel() // c
    ostride = 0
    if offset is not None:
        if offset.dim() == 1:
            if offset.shape[0] != c or not offset.is_contiguous():
                raise ValueError(f"per-channel offset must be a contiguous [{c}] tensor")
        else:
            if tuple(offset.shape) != tuple(loc.shape) or offset.stride(-1) != 1:
                raise ValueError(f"offset shape {tuple(offset.shape)} does not match the latent {tuple(loc.shape)}")
            ostride = int(offset.stride(-2))
            n, h, w, _ = loc.shape
            if tuple(offset.stride()) != (h * w * ostride, w * ostride, ostride, 1):
                raise ValueError("offset must be dense over pixels (a contiguous tensor or a last-axis slice of one)")
        if offset.dtype != torch.float32 or not offset.is_cuda:
            raise ValueError("offset must be a float32 CUDA tensor")
    out = torch.empty_like(loc)
    capi.call("sntc_uq_sample", _ptr(loc), _ptr(offset), npix, c, ostride, UQ_MODES[mode], float(param), _ptr(noise),
              int(seed), int(step), _ptr(out), _stream())
    return out


def distortion_grad(x, x_hat, scale):
    """-> (g_xhat with x_hat's (padded) shape, sse[n] float64 of 255 (x - x_hat) over the un-padded region)."""
    _check_nhwc(x)
    _check_nhwc(x_hat, x.shape[-1])
    n, h, w, c = x.shape
    g = torch.empty_like(x_hat)
    sse = torch.empty((n,), dtype=torch.float64, device=x.device)
    capi.call("sntc_distortion_grad", _ptr(x), _ptr(x_hat), n, h, w, c, x_hat.shape[1], x_hat.shape[2], float(scale),
              _ptr(g), _ptr(sse), _stream())
    return g, sse


def two_layer_tail_bwd(t, g_h, ch, has_res, act_kind, beta, gamma, cp, param_operands=False):
    """-> g_t, or (g_t, |base|, g_h * base) when ``param_operands`` (the IGDN1 parameter-gradient operands)."""
    _check_nhwc(t, ch * (2 if has_res else 1))
    _check_nhwc(g_h, ch)
    n, hh, wh, _ = t.shape
    g_t = torch.empty((n, hh, wh, cp), dtype=torch.float32, device=t.device)
    ax = torch.empty((n, hh, wh, ch), dtype=torch.float32, device=t.device) if param_operands else None
    gx = torch.empty_like(ax) if param_operands else None
    capi.call("sntc_two_layer_tail_bwd", _ptr(t), _ptr(g_h), n * hh * wh, ch, int(has_res), act_kind, _ptr(beta), _ptr(gamma),
              cp, _ptr(g_t), _ptr(ax), _ptr(gx), _stream())
    return (g_t, ax, gx) if param_operands else g_t


def two_layer_out_adjoint(g_xhat, w2, ch, k2=5, s2=2):
    """Gradient w.r.t. the hidden layer of a two-layer decoder from the gradient w.r.t. its output image."""
    _check_nhwc(g_xhat, 3)
    n, H, W, _ = g_xhat.shape
    g_h = torch.empty((n, H // s2, W // s2, ch), dtype=torch.float32, device=g_xhat.device)
    capi.call("sntc_two_layer_out_adjoint", _ptr(g_xhat), n, H // s2, W // s2, ch, _ptr(w2), k2, s2, 3, _ptr(g_h), _stream())
    return g_h


def adam_step(param, grad, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    """In-place Keras Adam update of ``param`` (and its moments m, v); t is the 1-based step."""
    capi.call("sntc_adam_step", _ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), float(lr), float(beta1),
              float(beta2), float(eps), int(t), float(grad_scale), _stream())


# ------------------------------------------------------------------------------------------
# Training step pieces
# ------------------------------------------------------------------------------------------

_WS = {}


def _workspace(nbytes, device):
    """One growing scratch buffer per (device, launch stream) for the split-K slabs of the gradient kernels: launches on
    one stream are ordered, launches on different streams (the trainer's weight-gradient stream) must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(int(nbytes), 1 << 20),), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def conv_wgrad(kind, k, stride, cin, cout, x, g_out, dw, accumulate=False):
    """dw (the layer's kernel layout, flat or shaped) = or += d loss / d kernel; g_out is the pre-activation gradient."""
    _check_nhwc(x, cin)
    _check_nhwc(g_out, cout)
    n, h, w, _ = x.shape
    kid = KINDS[kind]
    need = capi.load().sntc_conv_wgrad_workspace_bytes(kid, k, k, stride, cin, cout, n, h, w)
    if need < 0:
        raise capi.SntcError(capi.ERR_UNSUPPORTED, capi.last_error())
    ws = _workspace(need, x.device)
    capi.call("sntc_conv_wgrad", kid, k, k, stride, cin, cout, _ptr(x), _ptr(g_out), n, h, w, _ptr(dw), int(accumulate), _ptr(ws),
              ws.numel(), _stream())


def bias_grad(g, db, accumulate=False):
    c = g.shape[-1]
    npix = g.numel() // c
    need = capi.load().sntc_bias_grad_workspace_bytes(npix, c)
    ws = _workspace(need, g.device)
    capi.call("sntc_bias_grad", _ptr(g), npix, c, _ptr(db), int(accumulate), _ptr(ws), ws.numel(), _stream())


def act_backward(g, y, act, out=None):
    out = torch.empty_like(g) if out is None else out
    capi.call("sntc_act_backward", _ptr(g), _ptr(y), g.numel(), ACTS[act], _ptr(out), _stream())
    return out


def gate_forward(x, t, s):
    out = torch.empty_like(x)
    capi.call("sntc_gate_forward", _ptr(x), _ptr(t), _ptr(s), x.numel(), _ptr(out), _stream())
    return out


def gate_backward(g, t, s):
    g_t, g_s = torch.empty_like(g), torch.empty_like(g)
    capi.call("sntc_gate_backward", _ptr(g), _ptr(t), _ptr(s), g.numel(), _ptr(g_t), _ptr(g_s), _stream())
    return g_t, g_s


def axpy(a, b, alpha=1.0):
    capi.call("sntc_axpy", _ptr(a), _ptr(b), float(alpha), a.numel(), _stream())
    return a


def noise_add(x, noise=None, seed=0, step=0):
    out = torch.empty_like(x)
    capi.call("sntc_noise_add", _ptr(x), x.numel(), _ptr(noise), int(seed), int(step), _ptr(out), _stream())
    return out


def sumsq(x):
    out = torch.empty((1,), dtype=torch.float64, device=x.device)
    capi.call("sntc_sumsq", _ptr(x), x.numel(), _ptr(out), _stream())
    return out


def gdn_apply(x, norm, inverse):
    y = torch.empty_like(x)
    capi.call("sntc_gdn_apply", _ptr(x), _ptr(norm), x.numel(), int(inverse), _ptr(y), _stream())
    return y


def gdn_backward_prep(g, x, norm, inverse):
    """-> (q = d loss / d norm, |x|)"""
    q, ax = torch.empty_like(x), torch.empty_like(x)
    capi.call("sntc_gdn_backward_prep", _ptr(g), _ptr(x), _ptr(norm), x.numel(), int(inverse), _ptr(q), _ptr(ax), _stream())
    return q, ax


def gdn_backward_finish(g, x, norm, t, inverse):
    dx = torch.empty_like(x)
    capi.call("sntc_gdn_backward_finish", _ptr(g), _ptr(x), _ptr(norm), _ptr(t), x.numel(), int(inverse), _ptr(dx), _stream())
    return dx


def small_matmul(a, b, out, transpose_a=False):
    """out[rows, cols] = a @ b (transpose_a: a.T @ b) for a small ``a``; all 2-D contiguous float32 device tensors."""
    k, cols = int(b.shape[0]), int(b.shape[1])
    rows = int(a.shape[1] if transpose_a else a.shape[0])
    capi.call("sntc_small_matmul", _ptr(a), _ptr(b), rows, k, cols, int(transpose_a), _ptr(out), _stream())
    return out


def transpose_last2(src, dst):
    """dst[t, b, a] = src[t, a, b] for 3-D views of contiguous tensors."""
    t, a, b = (int(v) for v in src.shape)
    capi.call("sntc_transpose_last2", _ptr(src), t, a, b, _ptr(dst), _stream())
    return dst


def two_layer_hidden(t, ch, has_res, act_kind, beta, gamma):
    n, hh, wh, _ = t.shape
    h = torch.empty((n, hh, wh, ch), dtype=torch.float32, device=t.device)
    capi.call("sntc_two_layer_hidden", _ptr(t), n * hh * wh, ch, int(has_res), act_kind, _ptr(beta), _ptr(gamma), _ptr(h), _stream())
    return h


def noisy_normal(y_tilde, hyper):
    """-> (bits[n] float64, d bits / d (y~ - mu), d bits / d raw)."""
    n, h, w, c = y_tilde.shape
    dv, dr = torch.empty_like(y_tilde), torch.empty_like(y_tilde)
    bits = torch.empty((n,), dtype=torch.float64, device=y_tilde.device)
    capi.call("sntc_noisy_normal", _ptr(y_tilde), _ptr(hyper), n, h * w, c, _ptr(dv), _ptr(dr), _ptr(bits), _stream())
    return bits, dv, dr


def noisy_factorized(prior, z_tilde):
    """bits[n] (float64) of explicit samples under the noisy deep-factorized density (the training=True value of
    ContinuousBatchedEntropyModel / ``prior.log_prob``, reference mshyper/models.py:253-268), and d bits / d z~."""
    _check_nhwc(z_tilde, prior.channels)
    n, hw = z_tilde.shape[0], z_tilde.shape[1] * z_tilde.shape[2]
    dbits = torch.empty_like(z_tilde)
    bits = torch.empty((n,), dtype=torch.float64, device=z_tilde.device)
    capi.call("sntc_noisy_factorized", prior._h, _ptr(z_tilde), n, hw, _ptr(dbits), _ptr(None), _ptr(bits), _stream())
    return bits, dbits


# ------------------------------------------------------------------------------------------
# SSIM / MS-SSIM (eval-only quality metrics)
# ------------------------------------------------------------------------------------------
MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def pixels_float(x_hat, h, w):
    """crop + (v+.5)*255 + round-half-even + clamp, as float32 [n,h,w,c]."""
    _check_nhwc(x_hat)
    n, hs, ws, c = x_hat.shape
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x_hat.device)
    capi.call("sntc_pixels_float", _ptr(x_hat), n, h, w, c, hs, ws, _ptr(out), _stream())
    return out


def _ssim_scale(a, b, max_val, out):
    """out: float64 [2, n, c] slice that receives the spatial sums of luminance*cs and of cs."""
    n, h, w, c = a.shape
    capi.call("sntc_ssim_scale", _ptr(a), _ptr(b), n, h, w, c, float(max_val), _ptr(out[0]), _ptr(out[1]), _stream())
    return float((h - 10) * (w - 10))


def _avgpool2(x):
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float32, device=x.device)
    capi.call("sntc_avgpool2_symmetric", _ptr(x), n, h, w, c, _ptr(y), _stream())
    return y


def image_quality_launch(a, b, max_val=255.0):
    """Launch part of ``image_quality``: -> (device sums [scales, 2, n, c] float64, per-scale window counts, single-scale flag).
    Nothing is copied to the host, so several images can be in flight (Model.evaluate's look-ahead)."""
    _check_nhwc(a)
    _check_nhwc(b, a.shape[-1])
    n, h, w, c = a.shape
    single = h < 160 and w < 160
    scales = 1 if single else len(MSSSIM_WEIGHTS)
    sums = torch.empty((scales, 2, n, c), dtype=torch.float64, device=a.device)
    counts = []
    for k in range(scales):
        if k > 0:
            a, b = _avgpool2(a), _avgpool2(b)
        counts.append(_ssim_scale(a, b, max_val, sums[k]))
    return sums, counts, single


def image_quality_finish(sums_host, counts, single):
    """Host part: the per-(image, channel) sums of every scale -> per-image (MS-)SSIM (float64 array)."""
    means = np.asarray(sums_host) / np.asarray(counts).reshape(-1, 1, 1, 1)        # [scales, 2, n, c]
    if single:
        return means[0, 0].mean(axis=-1)
    scales = means.shape[0]
    factors = [np.maximum(means[k, 1], 0.0) for k in range(scales - 1)] + [np.maximum(means[-1, 0], 0.0)]
    stack = np.stack(factors, axis=-1)                                          # [n, c, scales]
    return np.prod(stack ** np.asarray(MSSSIM_WEIGHTS), axis=-1).mean(axis=-1)


def image_quality(a, b, max_val=255.0):
    """reference mshyper/models.py:321-331 on pixel-valued float images [n,h,w,c]: tf.image.ssim when both
    sides are < 160, tf.image.ssim_multiscale otherwise.  -> per-image msssim as a float64 host array.
    The kernels leave per-(image, channel) sums; the 5-factor geometric mean is finished on the host."""
    sums, counts, single = image_quality_launch(a, b, max_val)
    return image_quality_finish(sums.cpu().numpy(), counts, single)
