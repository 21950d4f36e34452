"""Analysis / synthesis / hyper transforms: the reference's registry, MI355X arithmetic.

Same class names, keyword arguments and call convention as reference common/transforms.py
(registry :380-393) and common/elic.py; ``class_builder.build(cls_name, **kwargs)`` is the plugin
point mshyper/models.py:112-131 uses.  A transform is ``t(x, training=False)`` on NHWC float32
CUDA tensors.  Like a Keras layer it is *built* on first use (input channels are inferred from the
input) with framework-default initial values -- glorot-uniform kernels, zero biases, GDN beta = 1,
gamma = 0.1 I -- and ``set_weights`` loads trained values given as ``{name: ndarray}`` in the
Keras / TFC variable layouts (Conv2D [kh,kw,Cin,Cout]; Conv2DTranspose [kh,kw,Cout,Cin];
SignalConv2D [kh,kw,Cin,Cout]; GDN beta [C], gamma [Cin,Cout], effective values).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .. import ops
from ._graph import GDN, Conv, DualPlan, ResidualBlock, Seq, SimpleAttention, set_precision
from .utils import ClassBuilder


def get_activation_op(activation, name="act"):
    """reference transforms.py:66-78 -> (conv epilogue activation, separate GDN node or None)."""
    if activation is None:
        return None, None
    if activation == "prelu":
        # tf.keras.layers.PReLU() with default shared_axes=None owns one alpha per (H, W, C) POSITION of its first input: such a
        # model only runs at the one resolution it was built at; no reference config uses it (transforms.py:69-70)
        raise NotImplementedError("prelu: Keras' PReLU has one alpha per spatial position (a fixed-resolution model); not supported")
    a = activation.lower()
    if a in ("gdn", "gdn1"):
        return None, GDN(name, inverse=False)
    if a in ("igdn", "igdn1"):
        return None, GDN(name, inverse=True)
    if a == "lrelu":
        a = "leaky_relu"
    if a not in ("relu", "leaky_relu", "sigmoid"):
        raise NotImplementedError(f"activation {activation!r}")
    return a, None


def _glorot(rng, shape):
    rec = shape[0] * shape[1]
    limit = np.sqrt(6.0 / (shape[2] * rec + shape[3] * rec))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def default_init(shapes, seed):
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shp in shapes.items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf == "kernel":
            out[name] = _glorot(rng, shp)
        elif leaf == "bias":
            out[name] = np.zeros(shp, np.float32)
        elif leaf == "beta":
            out[name] = np.ones(shp, np.float32)
        elif leaf == "gamma":
            out[name] = (0.1 * np.eye(shp[0])).astype(np.float32)
        else:
            raise KeyError(name)
    return out


class Transform:
    """Base: holds the node graph, host weights, and (once built) the packed device plans."""

    def __init__(self, graph, input_channels=None, seed=4321):
        self._graph = graph
        self._cin = input_channels
        self._seed = seed
        self._precision = "fp32"      # "bf16x3": the convolutions that qualify run split precision (Model(precision=...))
        self._weights = None          # host, OrderedDict name -> float32 ndarray
        self._built_on = None
        self.output_channels = None

    # -- variables ---------------------------------------------------------------------------
    def param_shapes(self, input_channels=None):
        cin = input_channels or self._cin
        if cin is None:
            raise ValueError("input channels unknown: pass input_channels or call the transform once")
        return self._graph.shapes(cin)[0]

    def num_params(self, input_channels=None):
        return int(sum(int(np.prod(s)) for s in self.param_shapes(input_channels).values()))

    def get_weights(self):
        if self._weights is None:
            self._weights = default_init(self.param_shapes(), self._seed)
        return self._weights

    def set_weights(self, weights, input_channels=None):
        if input_channels is not None:
            self._cin = input_channels
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in weights]
        if missing:
            raise KeyError(f"missing variables: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        w = OrderedDict()
        for k, shp in shapes.items():
            a = np.asarray(weights[k], dtype=np.float32)
            if tuple(a.shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {tuple(shp)}, got {tuple(a.shape)}")
            w[k] = a
        self._weights = w
        self._built_on = None

    def build(self, input_channels=None, device=None):
        if input_channels is not None:
            if self._cin is not None and self._cin != input_channels and self._built_on is not None:
                raise ValueError(f"transform built for {self._cin} input channels, got {input_channels}")
            self._cin = input_channels
        device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self._built_on == device:
            return self
        host = self._prepare_weights(self.get_weights())
        dev = {k: ops.to_device(v, device) for k, v in host.items()}
        set_precision(self._graph, self._precision)
        with torch.cuda.device(device):
            self.output_channels = self._graph.build(dev, self._graph_cin())
        self._dev = dev
        self._built_on = device
        return self

    def _graph_cin(self):
        return self._cin

    def _prepare_weights(self, w):
        return w

    # -- call -----------------------------------------------------------------------------------
    def __call__(self, x, training=False):
        if self._built_on != x.device:
            self.build(x.shape[-1] if x.dim() == 4 else x.shape[3] * 16, x.device)
        return self._forward(x)

    def _forward(self, x):
        return self._graph(x)

    def out_channels(self, input_channels=None):
        return self._graph.shapes(input_channels or self._cin)[1]

    def out_hw(self, h, w):
        """Output spatial size for an h x w input (shape inference only, nothing is launched)."""
        return self._graph.out_hw(h, w)

    def takes_s3(self, h, w):
        """True if the transform's FIRST convolution reads pre-split (format S3) input for h x w images: its producer can then
        emit S3 directly (the decoder's dequantisation does, ops.dequant_split3)."""
        first = self._graph.layers[0] if isinstance(self._graph, Seq) and self._graph.layers else None
        plan = getattr(first, "plan", None)
        return isinstance(plan, DualPlan) and plan.takes_s3(h, w)


class BLS2017Analysis(Transform):
    """reference transforms.py:93-112."""

    def __init__(self, num_filters):
        f = num_filters
        super().__init__(Seq([Conv("layer_0", "sigdown", f, 9, 4), GDN("gdn_0"),
                              Conv("layer_1", "sigdown", f, 5, 2), GDN("gdn_1"),
                              Conv("layer_2", "sigdown", f, 5, 2, bias=False)]), 3)


class BLS2017Synthesis(Transform):
    """reference transforms.py:115-134."""

    def __init__(self, num_filters):
        f = num_filters
        super().__init__(Seq([Conv("layer_0", "sigup", f, 5, 2), GDN("igdn_0", inverse=True),
                              Conv("layer_1", "sigup", f, 5, 2), GDN("igdn_1", inverse=True),
                              Conv("layer_2", "sigup", 3, 9, 4)]))


class MBT2018Analysis(Transform):
    """reference transforms.py:137-155.  ``tfc.GDN(name=...)`` is constructed with TFC's defaults;
    gdn_alpha / gdn_epsilon choose the form (1, 1 = TFC 2.x default; 2, 0.5 = classic GDN)."""

    def __init__(self, channels_base, n_layers=4, output_channels=None, gdn_alpha=1, gdn_epsilon=1.0):
        layers = []
        for i in range(n_layers):
            last = i + 1 == n_layers
            ch = (output_channels if output_channels is not None else channels_base) if last else channels_base
            layers.append(Conv(f"layer_{i}", "sigdown", ch, 5, 2))
            if not last:
                layers.append(GDN(f"gdn_{i}", False, gdn_alpha, gdn_epsilon))
        super().__init__(Seq(layers), 3)


class MBT2018Synthesis(Transform):
    """reference transforms.py:158-175."""

    def __init__(self, channels_base, n_layers=4, output_channels=3, gdn_alpha=1, gdn_epsilon=1.0):
        layers = []
        for i in range(n_layers):
            last = i + 1 == n_layers
            ch = (output_channels if output_channels is not None else channels_base) if last else channels_base
            layers.append(Conv(f"layer_{i}", "sigup", ch, 5, 2))
            if not last:
                layers.append(GDN(f"igdn_{i}", True, gdn_alpha, gdn_epsilon))
        super().__init__(Seq(layers))


def _four_layer(kind, channels_base, last_channels, activation_type):
    act, extra = get_activation_op(activation_type)     # ONE activation object shared by the layers (:183,199)
    layers = []
    for i in range(4):
        last = i == 3
        layers.append(Conv(f"layer_{i}", kind, last_channels if last else channels_base, 5, 2, None if last else act))
        if extra is not None and not last:
            layers.append(extra)
    return Seq(layers)


class CNNAnalysis(Transform):
    """reference transforms.py:179-192."""

    def __init__(self, channels_base, output_channels=None, activation_type="leaky_relu"):
        oc = channels_base if output_channels is None else output_channels
        super().__init__(_four_layer("conv", channels_base, oc, activation_type), 3)


class CNNSynthesis(Transform):
    """reference transforms.py:195-206."""

    def __init__(self, channels_base, output_channels=3, activation_type="leaky_relu"):
        super().__init__(_four_layer("convT", channels_base, output_channels, activation_type))


class HyperAnalysis(Transform):
    """reference transforms.py:209-219."""

    def __init__(self, bottleneck_size, activation_type="relu"):
        act, _ = get_activation_op(activation_type)
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "conv", b, 3, 1, act), Conv("layer_1", "conv", b, 5, 2, act),
                              Conv("layer_2", "conv", b, 5, 2, None)]))


class HyperSynthesis(Transform):
    """reference transforms.py:222-232."""

    def __init__(self, bottleneck_size, activation_type="relu"):
        act, _ = get_activation_op(activation_type)
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "convT", b, 5, 2, act), Conv("layer_1", "convT", int(b * 1.5), 5, 2, act),
                              Conv("layer_2", "convT", b * 2, 3, 1, None)]))


class HyperAnalysisSmall(Transform):
    """reference transforms.py:235-247."""

    def __init__(self, bottleneck_size):
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "sigdown", b, 3, 1, "relu"),
                              Conv("layer_1", "sigdown", b, 5, 2, None, bias=False)]))


class HyperSynthesisSmall(Transform):
    """reference transforms.py:250-262."""

    def __init__(self, bottleneck_size):
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "sigup", int(b * 1.5), 5, 2, "relu"),
                              Conv("layer_1", "sigup", int(b * 2), 3, 1, None)]))


class JPEGLikeSynthesis(Transform):
    """reference transforms.py:265-295: one Conv2DTranspose(k, s) mapping each latent vector to an
    overlapping k x k x 3 patch.  ``use_offset`` (:291-293) appends a channel of ones to the input: the kernel variable then
    has Cin + 1 input channels; the device plan carries it padded to Cin + 16 (zero weights) so that the layer stays on the
    16-channel vector loader, and the input gets the ones channel and fifteen zero channels appended per call."""

    def __init__(self, output_channels=3, kernel_size=16, strides=16, padding="SAME", use_bias=True, use_offset=False):
        if padding != "SAME":
            raise NotImplementedError("only padding='SAME' is used by the reference configs")
        self._use_offset = bool(use_offset)
        super().__init__(Seq([Conv("conv", "convT", output_channels, kernel_size, strides, None, use_bias)]))

    def param_shapes(self, input_channels=None):
        cin = input_channels or self._cin
        if cin is None:
            raise ValueError("input channels unknown: pass input_channels or call the transform once")
        return self._graph.shapes(cin + 1 if self._use_offset else cin)[0]

    def _graph_cin(self):
        return self._cin + 16 if self._use_offset else self._cin

    def _prepare_weights(self, w):
        if not self._use_offset:
            return w
        w = OrderedDict(w)
        k = w["conv/kernel"]                                  # Keras transposed layout [kh, kw, Cout, Cin + 1]
        w["conv/kernel"] = np.concatenate([k, np.zeros(k.shape[:3] + (15,), np.float32)], axis=3)
        return w

    def takes_s3(self, h, w):
        return not self._use_offset and super().takes_s3(h, w)

    def _forward(self, x):
        if self._use_offset:
            x = ops.concat_channels(x, None, 16)
        return self._graph(x)


class JPEGLikeHyperSynthesis(Transform):
    """reference transforms.py:364-377."""

    def __init__(self, bottleneck_size, kernel_size=6):
        super().__init__(Seq([Conv("conv", "convT", bottleneck_size * 2, kernel_size, 4, None)]))


class _TwoLayerBase(Transform):
    """Shared driver of the two-layer syntheses.  The shapes the reference's configs use (hidden width 12 / 24 / 48, a 5x5
    stride-2 output layer with 3 channels, a convolutional residual) run as ONE stride-8 transposed conv producing
    [base | res] (the two 13x13 kernels are concatenated along Cout, since they read the same input) followed by the fused
    tail kernel (activation + residual add + output layer [+ crop + uint8 + SSE]).  Every other registered shape -- another
    hidden width or output layer, res_type="d2s" (reference transforms.py:339-348) -- runs layer by layer on the generic
    plans: base conv (+ relu / leaky-relu in its epilogue), GDN / IGDN, residual branch, add, output conv."""

    def __init__(self, channels, strides, kernel_sizes, activation_type, has_res, res_type="conv"):
        self._ch, self._out_ch = int(channels[0]), int(channels[1])
        self._s, self._k = tuple(strides), tuple(kernel_sizes)
        self._has_res = has_res
        if res_type not in ("conv", "d2s"):
            raise NotImplementedError(f"res_type {res_type!r}")           # as the reference (transforms.py:349-350)
        self._res_type = res_type if has_res else None
        a = None if activation_type is None else activation_type.lower()
        if a not in ops.TAIL_ACTS:
            raise NotImplementedError(f"activation {activation_type!r} in the two-layer synthesis")
        self._act_name = {"lrelu": "leaky_relu", "none": None}.get(a, a)
        self._act_kind = ops.TAIL_ACTS[a]
        self._fused = (self._ch in ops.TAIL_CHANNELS and self._k[1] == 5 and self._s[1] == 2 and self._out_ch == 3)
        self._merged = self._fused and self._res_type != "d2s"     # [base | res] from one launch, then the fused tail
        if self._res_type == "d2s" and self._s[0] != 8:
            raise ValueError("res_type='d2s' upsamples by 2 x 2 x 2: the first layer's stride must be 8")
        super().__init__(None)
        self._syn = None                  # the fused first-layer plan, created by build()
        self._names = ("base_conv", "res", "out_conv") if has_res else ("conv1", None, "conv2")

    def param_shapes(self, input_channels=None):
        cin = input_channels or self._cin
        if cin is None:
            raise ValueError("input channels unknown")
        n1, nr, n2 = self._names
        d = OrderedDict()
        d[f"{n1}/kernel"] = (self._k[0], self._k[0], self._ch, cin)
        d[f"{n1}/bias"] = (self._ch,)
        if self._act_kind in (1, 2):
            d["act/beta"] = (self._ch,)
            d["act/gamma"] = (self._ch, self._ch)
        if self._res_type == "conv":
            d[f"{nr}/kernel"] = (self._k[0], self._k[0], self._ch, cin)
            d[f"{nr}/bias"] = (self._ch,)
        elif self._res_type == "d2s":
            if cin % 4 or (192 % 4) or (self._ch * 4) % 4:
                raise ValueError(f"res_type='d2s' needs input channels divisible by 4, got {cin}")
            d[f"{nr}/conv0/kernel"] = (1, 1, cin // 4, 192)               # Keras names them conv2d_<n>; ours: res/conv0, res/conv1
            d[f"{nr}/conv0/bias"] = (192,)
            d[f"{nr}/conv1/kernel"] = (1, 1, 48, self._ch * 4)
            d[f"{nr}/conv1/bias"] = (self._ch * 4,)
        d[f"{n2}/kernel"] = (self._k[1], self._k[1], self._out_ch, self._ch)
        d[f"{n2}/bias"] = (self._out_ch,)
        return d

    def out_channels(self, input_channels=None):
        return self._out_ch

    def out_hw(self, h, w):
        return h * self._s[0] * self._s[1], w * self._s[0] * self._s[1]

    def build(self, input_channels=None, device=None):
        if input_channels is not None:
            self._cin = input_channels
        device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self._built_on == device:
            return self
        w = self.get_weights()
        n1, nr, n2 = self._names
        dv = lambda a: ops.to_device(a, device)
        k1, b1 = w[f"{n1}/kernel"], w[f"{n1}/bias"]
        merged = self._merged
        if merged and nr:
            k1 = np.concatenate([k1, w[f"{nr}/kernel"]], axis=2)
            b1 = np.concatenate([b1, w[f"{nr}/bias"]])
        with torch.cuda.device(device):
            epi_act = self._act_name if (not self._fused and self._act_kind in (3, 4)) else None     # else the tail / GDN node applies it
            self._up = DualPlan("convT", dv(k1), dv(b1), self._s[0], epi_act, precision=self._precision)
            self._res_plan = self._d2s = self._out_plan = None
            if self._res_type == "conv" and not merged:
                self._res_plan = ops.ConvPlan("convT", dv(w[f"{nr}/kernel"]), dv(w[f"{nr}/bias"]), self._s[0])
            if self._res_type == "d2s":
                self._d2s = (ops.ConvPlan("conv", dv(w[f"{nr}/conv0/kernel"]), dv(w[f"{nr}/conv0/bias"]), 1, "leaky_relu"),
                             ops.ConvPlan("conv", dv(w[f"{nr}/conv1/kernel"]), dv(w[f"{nr}/conv1/bias"]), 1, "leaky_relu"))
            if not self._fused:
                self._out_plan = ops.ConvPlan("convT", dv(w[f"{n2}/kernel"]), dv(w[f"{n2}/bias"]), self._s[1])
            self._gdn_plan = None
            if self._act_kind in (1, 2) and not self._fused and self._ch not in ops.GDN_SMALL_CHANNELS:
                from .. import _capi as capi
                epi = capi.EPI_RES_MUL if self._act_kind == 1 else capi.EPI_RES_DIV
                self._gdn_plan = ops.ConvPlan("conv", dv(w["act/gamma"].reshape(1, 1, self._ch, self._ch)), dv(w["act/beta"]), 1, None,
                                              capi.PRO_ABS, epi)
        self._beta = dv(w["act/beta"]) if "act/beta" in w else None
        self._gamma = dv(w["act/gamma"]) if "act/gamma" in w else None
        self._w2 = dv(w[f"{n2}/kernel"])
        self._b2 = dv(w[f"{n2}/bias"])
        # the first layer, its activation and the residual add in ONE launch (csrc/syn_fused.hip) where the kernel exists:
        # [base | res] never goes to HBM, the tail kernel then only runs the output layer on the hidden tensor.  fp32 only
        # (the split-precision mode keeps the pre-split gather GEMM); same bits as the layers.
        self._syn = None
        if merged and self._precision == "fp32" and ops.SynPlan.supported(self._k[0], self._s[0], self._cin, self._ch, self._has_res):
            with torch.cuda.device(device):
                self._syn = ops.SynPlan(dv(k1), dv(b1), self._s[0], self._ch, self._has_res, self._act_kind, self._beta, self._gamma)
        self.output_channels = self._out_ch
        self._built_on = device
        return self

    def takes_s3(self, h, w):
        return self._built_on is not None and self._merged and self._up.takes_s3(h, w)

    def _res_d2s(self, x):
        c0, c1 = self._d2s
        return ops.depth_to_space(c1(ops.depth_to_space(c0(ops.depth_to_space(x)))))

    def _hidden(self, x):
        """Layer by layer: act(base_conv(x)) + res(x), fp32 [n, 8h, 8w, ch]."""
        base = self._up(x)                                        # relu / leaky-relu already applied in its epilogue
        if self._act_kind in (1, 2):
            if self._gdn_plan is not None:
                base = self._gdn_plan(base, res=base)
            else:
                base = ops.gdn_small(base, self._beta, self._gamma, inverse=self._act_kind == 1)
        if self._res_type == "conv":
            base = ops.axpy(base, self._res_plan(x))
        elif self._res_type == "d2s":
            base = ops.axpy(base, self._res_d2s(x))
        return base

    def _tail_input(self, x):
        """[base | res] for the fused tail: one launch, or (d2s) the two branches concatenated."""
        if self._merged:
            return self._up(x)
        return ops.concat_channels(self._up(x), self._res_d2s(x))

    def _ensure_built(self, x, cin=None):
        """(Re)build on x's device if the transform has never been built there or its weights were replaced since
        (Transform.set_weights clears _built_on): a direct caller must never run a plan packed from older weights."""
        if self._built_on != x.device:
            self.build(cin if cin is not None else (x.shape[-1] if x.dim() == 4 else x.shape[3] * 16), x.device)

    def _use_syn(self, x, alone=True):
        ok = (self._syn is not None and ops.FUSED_SYNTHESIS and x.dim() == 4 and self._syn.fits(x)
              and (2 if self._has_res else 1) * self._ch >= ops.FUSED_SYNTHESIS_MIN_COLUMNS)
        return ok and (not alone or self._syn.items([x]) >= ops.FUSED_SYNTHESIS_MIN_ITEMS)

    def hidden_many(self, xs):
        """act(base_conv(y_hat)) + res(y_hat) for a LIST of batches of different image sizes in one launch (None where the
        fused kernel does not apply: the caller then takes ``forward_pixels`` per batch)."""
        if not 1 <= len(xs) <= 4:
            return None
        self._ensure_built(xs[0])
        if not all(self._use_syn(x, alone=False) for x in xs) or self._syn.items(xs) < ops.FUSED_SYNTHESIS_MIN_ITEMS:
            return None
        return self._syn(list(xs))

    def pixels_from_hidden(self, hid, h, w, reference=None):
        """The output layer on the hidden tensor (+ crop + uint8 + SSE): the tail kernel without activation and residual."""
        self._ensure_built(hid, self._cin)
        return ops.two_layer_tail_pixels(hid, self._ch, False, 0, None, None, self._w2, self._b2, h, w, reference, self._k[1], self._s[1])

    def _forward(self, x):
        if self._fused and self._use_syn(x):
            return ops.two_layer_tail(self._syn(x), self._ch, False, 0, None, None, self._w2, self._b2, self._k[1], self._s[1])
        if self._fused:
            return ops.two_layer_tail(self._tail_input(x), self._ch, self._has_res, self._act_kind, self._beta, self._gamma,
                                      self._w2, self._b2, self._k[1], self._s[1])
        return self._out_plan(self._hidden(x))

    def forward_pixels(self, x, h, w, reference=None):
        """Decoder form: the synthesis ends in uint8 pixels cropped to h x w (and the integer SSE against ``reference``);
        for the fused shapes in the same launch as the activation and the output layer -- no float image round trip."""
        self._ensure_built(x)
        if self._fused and self._use_syn(x):
            return self.pixels_from_hidden(self._syn(x), h, w, reference)
        if self._fused:
            return ops.two_layer_tail_pixels(self._tail_input(x), self._ch, self._has_res, self._act_kind, self._beta, self._gamma,
                                             self._w2, self._b2, h, w, reference, self._k[1], self._s[1])
        recon = self._out_plan(self._hidden(x))
        if reference is None:
            return ops.to_pixels(recon, h, w), None
        sse, px = ops.pixels_sse(reference, recon, want_pixels=True)
        return px, sse


class TwoLayerSynthesis(_TwoLayerBase):
    """reference transforms.py:298-317."""

    def __init__(self, channels=(24, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn"):
        super().__init__(channels, strides, kernel_sizes, activation_type, has_res=False)


class TwoLayerResSynthesis(_TwoLayerBase):
    """reference transforms.py:320-361."""

    def __init__(self, channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn", res_type="conv"):
        super().__init__(channels, strides, kernel_sizes, activation_type, has_res=True, res_type=res_type)


class ElicAnalysis(Transform):
    """reference elic.py:103-177."""

    def __init__(self, num_residual_blocks=3, channels=(128, 160, 192, 192), kernel_sizes=(5, 5, 5, 5),
                 strides=(2, 2, 2, 2), output_channels=None, name="ElicAnalysis"):
        if len(channels) not in (3, 4):
            raise ValueError(f"ELIC uses 3 or 4 conv layers (not {channels}).")
        assert len(channels) == len(strides) == len(kernel_sizes)
        if output_channels is not None and output_channels != channels[-1]:
            raise ValueError(f"output_channels specified but does not match channels: {output_channels} vs. {channels}")
        self._downsample_factor = 2 ** len(channels)
        convs = [Conv(f"conv{i}", "conv", c, k, s) for i, (c, k, s) in enumerate(zip(channels, kernel_sizes, strides))]
        count = [0]

        def rbs():
            out = [ResidualBlock(f"rb{count[0] + j}") for j in range(num_residual_blocks)]
            count[0] += num_residual_blocks
            return out

        blocks = [convs[0], *rbs()] if len(channels) == 4 else []
        blocks += [convs[-3], *rbs(), SimpleAttention("attn0"), convs[-2], *rbs(), convs[-1], SimpleAttention("attn1")]
        super().__init__(Seq(blocks), 3)

    @property
    def output_depth(self):
        return self.out_channels()


class ElicSynthesis(Transform):
    """reference elic.py:180-250 (registered at transforms.py:389 but used by no shipped config)."""

    def __init__(self, num_residual_blocks=3, channels=(192, 160, 128, 3), kernel_sizes=(5, 5, 5, 5),
                 strides=(2, 2, 2, 2), output_channels=None, name="ElicSynthesis"):
        if len(channels) not in (3, 4):
            raise ValueError(f"ELIC uses 3 or 4 conv layers (not {channels}).")
        assert len(channels) == len(strides) == len(kernel_sizes)
        if output_channels is not None and output_channels != channels[-1]:
            raise ValueError(f"output_channels specified but does not match channels: {output_channels} vs. {channels}")
        convs = [Conv(f"conv{i}", "convT", c, k, s) for i, (c, k, s) in enumerate(zip(channels, kernel_sizes, strides))]
        count = [0]

        def rbs():
            out = [ResidualBlock(f"rb{count[0] + j}") for j in range(num_residual_blocks)]
            count[0] += num_residual_blocks
            return out

        blocks = [SimpleAttention("attn0"), convs[0], *rbs(), convs[1], SimpleAttention("attn1"), *rbs(), convs[2]]
        if len(channels) == 4:
            blocks += [*rbs(), convs[3]]
        super().__init__(Seq(blocks))


classes = [
    BLS2017Analysis, BLS2017Synthesis, CNNAnalysis, CNNSynthesis, HyperAnalysis, HyperSynthesis,
    MBT2018Analysis, MBT2018Synthesis, HyperAnalysisSmall, HyperSynthesisSmall, ElicAnalysis, ElicSynthesis,
    JPEGLikeSynthesis, TwoLayerSynthesis, TwoLayerResSynthesis, JPEGLikeHyperSynthesis,
]
# reference transforms.py:383-393
class_builder = ClassBuilder({cls.__name__: cls for cls in classes})
