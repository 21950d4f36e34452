"""float64 NumPy restatement of the reference's transform graphs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED at tensor level; the layer
inventory is pinned by the published parameter counts and FLOPs/pixel
(tests/test_oracle_pins.py).

Every class restates one class of reference common/transforms.py or common/elic.py under the
same name and keyword arguments.  A transform is  t(params, x) -> y  (NHWC float64) where
``params`` is a flat ``{name: ndarray}`` dict in the Keras / TFC variable layouts:

    Conv2D            kernel [kh,kw,Cin,Cout]  bias [Cout]
    Conv2DTranspose   kernel [kh,kw,Cout,Cin]  bias [Cout]
    SignalConv2D      kernel [kh,kw,Cin,Cout]  bias [Cout]      (both directions)
    GDN               beta [C]  gamma [C(in),C(out)]             (effective values)
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import ops_np as ops


# ---------------------------------------------------------------------------------------
# primitive layers
# ---------------------------------------------------------------------------------------
class _Layer:
    name = ""

    def shapes(self, cin):            # -> (OrderedDict name->shape, cout)
        raise NotImplementedError

    def flops(self, cin, h, w):       # -> (multiply-add*2 count, cout, h_out, w_out)
        raise NotImplementedError


class Conv(_Layer):
    """kind: 'conv' Keras Conv2D SAME | 'convT' Keras Conv2DTranspose SAME |
    'sigdown' tfc.SignalConv2D(corr=True, strides_down) | 'sigup' tfc.SignalConv2D(corr=False, strides_up)."""

    def __init__(self, name, kind, cout, k, s, act=None, bias=True):
        self.name, self.kind, self.cout, self.k, self.s, self.act, self.bias = name, kind, cout, k, s, act, bias

    def shapes(self, cin):
        d = OrderedDict()
        if self.kind == "convT":
            d[f"{self.name}/kernel"] = (self.k, self.k, self.cout, cin)
        else:
            d[f"{self.name}/kernel"] = (self.k, self.k, cin, self.cout)
        if self.bias:
            d[f"{self.name}/bias"] = (self.cout,)
        return d, self.cout

    def __call__(self, p, x, be=ops):
        w = p[f"{self.name}/kernel"]
        b = p.get(f"{self.name}/bias") if self.bias else None
        if self.kind == "conv":
            y = be.conv2d(x, w, b, self.s)
        elif self.kind == "convT":
            y = be.conv2d_transpose(x, w, b, self.s)
        elif self.kind == "sigdown":
            y = be.signal_conv_down(x, w, b, self.s)
        elif self.kind == "sigup":
            y = be.signal_conv_up(x, w, b, self.s)
        else:
            raise ValueError(self.kind)
        return be.ACTIVATIONS[self.act](y)

    def flops(self, cin, h, w):
        up = self.kind in ("convT", "sigup")
        ho, wo = (h * self.s, w * self.s) if up else (-(-h // self.s), -(-w // self.s))
        # dense 2*MAC count, the way the TF profiler counts conv / conv2d_backprop_input
        macs = (h * w if up else ho * wo) * self.k * self.k * cin * self.cout
        return 2 * macs, self.cout, ho, wo


class GDN(_Layer):
    def __init__(self, name, inverse=False, alpha=1, epsilon=1.0):
        self.name, self.inverse, self.alpha, self.epsilon = name, inverse, alpha, epsilon

    def shapes(self, cin):
        return OrderedDict([(f"{self.name}/beta", (cin,)), (f"{self.name}/gamma", (cin, cin))]), cin

    def __call__(self, p, x, be=ops):
        return be.gdn(x, p[f"{self.name}/beta"], p[f"{self.name}/gamma"], self.inverse,
                      self.alpha, self.epsilon)

    def flops(self, cin, h, w):
        return 2 * h * w * cin * cin, cin, h, w


class DepthToSpace(_Layer):
    """tf.nn.depth_to_space(x, 2) as a layer (common/transforms.py:342-347)."""

    def __init__(self, block=2):
        self.block = block

    def shapes(self, cin):
        assert cin % (self.block * self.block) == 0, cin
        return OrderedDict(), cin // (self.block * self.block)

    def __call__(self, p, x, be=ops):
        return be.depth_to_space(x, self.block)

    def flops(self, cin, h, w):
        return 0, cin // (self.block * self.block), h * self.block, w * self.block


class Seq(_Layer):
    def __init__(self, layers):
        self.layers = layers

    def shapes(self, cin):
        d = OrderedDict()
        for l in self.layers:
            s, cin = l.shapes(cin)
            for k, v in s.items():
                if k in d:                       # a shared activation object (transforms.py:183,199,306)
                    assert d[k] == v
                d[k] = v
        return d, cin

    def __call__(self, p, x, be=ops):
        for l in self.layers:
            x = l(p, x, be)
        return x

    def flops(self, cin, h, w):
        tot = 0
        for l in self.layers:
            f, cin, h, w = l.flops(cin, h, w)
            tot += f
        return tot, cin, h, w


class ResidualBlock(_Layer):
    """common/elic.py:41-68:  x + [1x1 c->c/2 relu, 3x3 c/2->c/2 relu, 1x1 c/2->c]."""

    def __init__(self, name):
        self.name = name

    def _block(self, c):
        n = self.name
        return Seq([Conv(f"{n}/conv0", "conv", c // 2, 1, 1, "relu"),
                    Conv(f"{n}/conv1", "conv", c // 2, 3, 1, "relu"),
                    Conv(f"{n}/conv2", "conv", c, 1, 1, None)])

    def shapes(self, cin):
        return self._block(cin).shapes(cin)

    def __call__(self, p, x, be=ops):
        return x + self._block(be.channels(x))(p, x, be)

    def flops(self, cin, h, w):
        return self._block(cin).flops(cin, h, w)


class SimpleAttention(_Layer):
    """common/elic.py:71-100:  x + trunk(x) * branch(x); trunk = 3 RB; branch = 3 RB + 1x1 sigmoid."""

    def __init__(self, name):
        self.name = name

    def _parts(self, c):
        n = self.name
        trunk = Seq([ResidualBlock(f"{n}/trunk/rb{i}") for i in range(3)])
        branch = Seq([ResidualBlock(f"{n}/branch/rb{i}") for i in range(3)]
                     + [Conv(f"{n}/branch/conv", "conv", c, 1, 1, "sigmoid")])
        return trunk, branch

    def shapes(self, cin):
        t, b = self._parts(cin)
        d, _ = t.shapes(cin)
        d2, _ = b.shapes(cin)
        d.update(d2)
        return d, cin

    def __call__(self, p, x, be=ops):
        t, b = self._parts(be.channels(x))
        return x + t(p, x, be) * b(p, x, be)

    def flops(self, cin, h, w):
        t, b = self._parts(cin)
        return t.flops(cin, h, w)[0] + b.flops(cin, h, w)[0], cin, h, w


# ---------------------------------------------------------------------------------------
# transforms (same names / kwargs as reference common/transforms.py:383-393)
# ---------------------------------------------------------------------------------------
class Transform:
    input_channels = None      # set by subclasses: 3 for analysis, bottleneck for synthesis ...

    def __init__(self, graph, cin):
        self.graph, self.cin = graph, cin

    def param_shapes(self):
        return self.graph.shapes(self.cin)[0]

    def num_params(self):
        return int(sum(np.prod(s) for s in self.param_shapes().values()))

    def flops(self, h, w):
        """2*MAC FLOPs for an input of spatial size h x w (conv / GDN contractions only)."""
        return self.graph.flops(self.cin, h, w)[0]

    def __call__(self, params, x, training=False, be=ops):
        return self.graph(params, be.as_input(x), be)


def _act_layers(activation_type, name):
    """transforms.py:66-78 get_activation_op: returns (conv_act, extra_layer)."""
    if activation_type is None:
        return None, None
    a = activation_type.lower()
    if a in ("gdn", "gdn1"):
        return None, GDN(name, inverse=False)
    if a in ("igdn", "igdn1"):
        return None, GDN(name, inverse=True)
    if a == "lrelu":
        a = "leaky_relu"
    return a, None


class ElicAnalysis(Transform):
    """common/elic.py:103-177."""

    def __init__(self, num_residual_blocks=3, channels=(128, 160, 192, 192), kernel_sizes=(5, 5, 5, 5),
                 strides=(2, 2, 2, 2), output_channels=None, cin=3):
        assert len(channels) in (3, 4) and len(channels) == len(strides) == len(kernel_sizes)
        convs = [Conv(f"conv{i}", "conv", c, k, s, None)
                 for i, (c, k, s) in enumerate(zip(channels, kernel_sizes, strides))]
        cnt = [0]

        def rbs():
            out = [ResidualBlock(f"rb{cnt[0] + j}") for j in range(num_residual_blocks)]
            cnt[0] += num_residual_blocks
            return out

        blocks = []
        if len(channels) == 4:
            blocks += [convs[0], *rbs()]
        blocks += [convs[-3], *rbs(), SimpleAttention("attn0"), convs[-2], *rbs(), convs[-1],
                   SimpleAttention("attn1")]
        super().__init__(Seq(blocks), cin)


class ElicSynthesis(Transform):
    """common/elic.py:180-250."""

    def __init__(self, num_residual_blocks=3, channels=(192, 160, 128, 3), kernel_sizes=(5, 5, 5, 5),
                 strides=(2, 2, 2, 2), output_channels=None, cin=None):
        assert len(channels) in (3, 4) and len(channels) == len(strides) == len(kernel_sizes)
        convs = [Conv(f"conv{i}", "convT", c, k, s, None)
                 for i, (c, k, s) in enumerate(zip(channels, kernel_sizes, strides))]
        cnt = [0]

        def rbs():
            out = [ResidualBlock(f"rb{cnt[0] + j}") for j in range(num_residual_blocks)]
            cnt[0] += num_residual_blocks
            return out

        blocks = [SimpleAttention("attn0"), convs[0], *rbs(), convs[1], SimpleAttention("attn1"), *rbs(), convs[2]]
        if len(channels) == 4:
            blocks += [*rbs(), convs[3]]
        super().__init__(Seq(blocks), cin)


class CNNAnalysis(Transform):
    """common/transforms.py:179-192 (one shared activation object, :183)."""

    def __init__(self, channels_base, output_channels=None, activation_type="leaky_relu", cin=3):
        output_channels = channels_base if output_channels is None else output_channels
        act, extra = _act_layers(activation_type, "act")
        layers = []
        for i in range(4):
            last = i == 3
            layers.append(Conv(f"layer_{i}", "conv", output_channels if last else channels_base, 5, 2,
                               None if last else act))
            if extra is not None and not last:
                layers.append(extra)
        super().__init__(Seq(layers), cin)


class CNNSynthesis(Transform):
    """common/transforms.py:195-206."""

    def __init__(self, channels_base, output_channels=3, activation_type="leaky_relu", cin=None):
        act, extra = _act_layers(activation_type, "act")
        layers = []
        for i in range(4):
            last = i == 3
            layers.append(Conv(f"layer_{i}", "convT", output_channels if last else channels_base, 5, 2,
                               None if last else act))
            if extra is not None and not last:
                layers.append(extra)
        super().__init__(Seq(layers), cin)


class HyperAnalysis(Transform):
    """common/transforms.py:209-219."""

    def __init__(self, bottleneck_size, activation_type="relu", cin=None):
        act, _ = _act_layers(activation_type, "act")
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "conv", b, 3, 1, act), Conv("layer_1", "conv", b, 5, 2, act),
                              Conv("layer_2", "conv", b, 5, 2, None)]), cin or b)


class HyperSynthesis(Transform):
    """common/transforms.py:222-232."""

    def __init__(self, bottleneck_size, activation_type="relu", cin=None):
        act, _ = _act_layers(activation_type, "act")
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "convT", b, 5, 2, act),
                              Conv("layer_1", "convT", int(b * 1.5), 5, 2, act),
                              Conv("layer_2", "convT", b * 2, 3, 1, None)]), cin or b)


class BLS2017Analysis(Transform):
    """common/transforms.py:93-112 (get_act -> GDN1)."""

    def __init__(self, num_filters, cin=3):
        f = num_filters
        super().__init__(Seq([Conv("layer_0", "sigdown", f, 9, 4), GDN("gdn_0"),
                              Conv("layer_1", "sigdown", f, 5, 2), GDN("gdn_1"),
                              Conv("layer_2", "sigdown", f, 5, 2, bias=False)]), cin)


class BLS2017Synthesis(Transform):
    """common/transforms.py:115-134."""

    def __init__(self, num_filters, cin=None):
        f = num_filters
        super().__init__(Seq([Conv("layer_0", "sigup", f, 5, 2), GDN("igdn_0", inverse=True),
                              Conv("layer_1", "sigup", f, 5, 2), GDN("igdn_1", inverse=True),
                              Conv("layer_2", "sigup", 3, 9, 4)]), cin or f)


class MBT2018Analysis(Transform):
    """common/transforms.py:137-155.  GDN is ``tfc.GDN(name=...)`` with TFC's constructor defaults;
    gdn_alpha/gdn_epsilon select them (TFC 2.x default alpha=1, epsilon=1; classic = 2, 0.5)."""

    def __init__(self, channels_base, n_layers=4, output_channels=None, cin=3, gdn_alpha=1, gdn_epsilon=1.0):
        layers = []
        for i in range(n_layers):
            last = i + 1 == n_layers
            ch = (output_channels if output_channels is not None else channels_base) if last else channels_base
            layers.append(Conv(f"layer_{i}", "sigdown", ch, 5, 2))
            if not last:
                layers.append(GDN(f"gdn_{i}", False, gdn_alpha, gdn_epsilon))
        super().__init__(Seq(layers), cin)


class MBT2018Synthesis(Transform):
    """common/transforms.py:158-175."""

    def __init__(self, channels_base, n_layers=4, output_channels=3, cin=None, gdn_alpha=1, gdn_epsilon=1.0):
        layers = []
        for i in range(n_layers):
            last = i + 1 == n_layers
            ch = (output_channels if output_channels is not None else channels_base) if last else channels_base
            layers.append(Conv(f"layer_{i}", "sigup", ch, 5, 2))
            if not last:
                layers.append(GDN(f"igdn_{i}", True, gdn_alpha, gdn_epsilon))
        super().__init__(Seq(layers), cin)


class HyperAnalysisSmall(Transform):
    """common/transforms.py:235-247."""

    def __init__(self, bottleneck_size, cin=None):
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "sigdown", b, 3, 1, "relu"),
                              Conv("layer_1", "sigdown", b, 5, 2, None, bias=False)]), cin or b)


class HyperSynthesisSmall(Transform):
    """common/transforms.py:250-262."""

    def __init__(self, bottleneck_size, cin=None):
        b = bottleneck_size
        super().__init__(Seq([Conv("layer_0", "sigup", int(b * 1.5), 5, 2, "relu"),
                              Conv("layer_1", "sigup", int(b * 2), 3, 1, None)]), cin or b)


class JPEGLikeSynthesis(Transform):
    """common/transforms.py:265-295."""

    def __init__(self, output_channels=3, kernel_size=16, strides=16, padding="SAME", use_bias=True,
                 use_offset=False, cin=None):
        assert padding == "SAME"
        self.use_offset = use_offset
        super().__init__(Seq([Conv("conv", "convT", output_channels, kernel_size, strides, None, use_bias)]),
                         (cin + 1) if use_offset else cin)

    def __call__(self, params, x, training=False, be=ops):
        x = be.as_input(x)
        if self.use_offset:
            x = be.append_ones(x)
        return self.graph(params, x, be)


class TwoLayerSynthesis(Transform):
    """common/transforms.py:298-317 (activation is conv1's Keras ``activation=``)."""

    def __init__(self, channels=(24, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn", cin=None):
        act, extra = _act_layers(activation_type, "act")
        layers = [Conv("conv1", "convT", channels[0], kernel_sizes[0], strides[0], act)]
        if extra is not None:
            layers.append(extra)
        layers.append(Conv("conv2", "convT", channels[1], kernel_sizes[1], strides[1], None))
        super().__init__(Seq(layers), cin)


class TwoLayerResSynthesis(Transform):
    """common/transforms.py:320-361:  out_conv(act(base_conv(z)) + res(z)); res is a second stride-8 transposed convolution
    (res_type='conv') or three depth_to_space steps with two 1x1 leaky-relu convolutions between them (res_type='d2s', :339-348;
    its variables are named res/conv0, res/conv1 here -- Keras would call them conv2d_<n>)."""

    def __init__(self, channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn",
                 res_type="conv", cin=None):
        act, extra = _act_layers(activation_type, "act")
        base = [Conv("base_conv", "convT", channels[0], kernel_sizes[0], strides[0], act)]
        if extra is not None:
            base.append(extra)
        self.base = Seq(base)
        if res_type == "conv":
            self.res = Conv("res", "convT", channels[0], kernel_sizes[0], strides[0], None)
        elif res_type == "d2s":
            self.res = Seq([DepthToSpace(), Conv("res/conv0", "conv", 192, 1, 1, "leaky_relu"), DepthToSpace(),
                            Conv("res/conv1", "conv", channels[0] * 4, 1, 1, "leaky_relu"), DepthToSpace()])
        else:
            raise NotImplementedError(res_type)
        self.out_conv = Conv("out_conv", "convT", channels[1], kernel_sizes[1], strides[1], None)
        self.cin = cin

    def param_shapes(self):
        d, c = self.base.shapes(self.cin)
        d.update(self.res.shapes(self.cin)[0])
        d.update(self.out_conv.shapes(c)[0])
        return d

    def flops(self, h, w):
        f0, c, ho, wo = self.base.flops(self.cin, h, w)
        f1 = self.res.flops(self.cin, h, w)[0]
        f2 = self.out_conv.flops(c, ho, wo)[0]
        return f0 + f1 + f2

    def __call__(self, params, x, training=False, be=ops):
        x = be.as_input(x)
        return self.out_conv(params, self.base(params, x, be) + self.res(params, x, be), be)


class JPEGLikeHyperSynthesis(Transform):
    """common/transforms.py:364-377."""

    def __init__(self, bottleneck_size, kernel_size=6, cin=None):
        super().__init__(Seq([Conv("conv", "convT", bottleneck_size * 2, kernel_size, 4, None)]),
                         cin or bottleneck_size)


CLASSES = {c.__name__: c for c in [
    BLS2017Analysis, BLS2017Synthesis, CNNAnalysis, CNNSynthesis, HyperAnalysis, HyperSynthesis,
    MBT2018Analysis, MBT2018Synthesis, HyperAnalysisSmall, HyperSynthesisSmall, ElicAnalysis, ElicSynthesis,
    JPEGLikeSynthesis, TwoLayerSynthesis, TwoLayerResSynthesis, JPEGLikeHyperSynthesis]}


def build(cls, **kwargs):
    """ClassBuilder.build (common/utils.py:58-71)."""
    return CLASSES[cls](**kwargs)


# ---------------------------------------------------------------------------------------
# synthetic weights (framework-default initialisers)
# ---------------------------------------------------------------------------------------
def init_params(shapes, rng, prefix=""):
    """kernel: glorot-uniform (Keras default); bias: zeros; beta: ones; gamma: 0.1*I (TFC default)."""
    out = OrderedDict()
    for name, shp in shapes.items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf == "kernel":
            rec = shp[0] * shp[1]
            limit = np.sqrt(6.0 / (shp[2] * rec + shp[3] * rec))
            v = rng.uniform(-limit, limit, size=shp)
        elif leaf == "bias":
            v = np.zeros(shp)
        elif leaf == "beta":
            v = np.ones(shp)
        elif leaf == "gamma":
            v = 0.1 * np.eye(shp[0])
        else:
            raise KeyError(name)
        out[prefix + name] = v.astype(np.float32)
    return out


def sub_params(params, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in params.items() if k.startswith(prefix)}
