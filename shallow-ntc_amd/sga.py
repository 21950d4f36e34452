"""SGA iterative inference: loss + gradients w.r.t. the latents (reference mshyper/models.py:389-413,
common/itinf_lib.py:26-93, common/latent_rvs_utils.py:8-48).

There is no autograd on this path: the backward pass is written out.  Its contractions are the same
gather-GEMM plans as the forward (the input gradient of a Keras Conv2DTranspose is the Conv2D with
the same kernel array), relu masks ride on the conv epilogue, everything else is element-wise HIP.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _capi as capi
from . import ops
from .common._graph import GDN, Conv, Seq
from .common.transforms import _TwoLayerBase

_MASKS = {None: None, "relu": capi.EPI_MASK_RELU, "leaky_relu": capi.EPI_MASK_LEAKY}


class GradChain:
    """Forward-with-cache and input-gradient of a stack of up-sampling layers: Keras Conv2DTranspose (HyperSynthesis,
    JPEGLikeSynthesis, CNNSynthesis with relu / leaky-relu) and tfc.SignalConv2D(strides_up) with inverse / forward GDN
    between them (MBT2018Synthesis, BLS2017Synthesis, HyperSynthesisSmall; GDN with alpha = 1, epsilon = 1).  The adjoint of an
    up layer is the down layer of the same kind on the same kernel array; the GDN input gradient is
    g / norm + sign(x) (q gamma^T) (inverse: g norm + ...), q = d loss / d norm, as in the training step (train.TGDN)."""

    def __init__(self, transform):
        graph = transform._graph
        if not isinstance(graph, Seq) or not all(isinstance(l, (Conv, GDN)) for l in graph.layers):
            raise NotImplementedError(f"SGA backward is implemented for Conv2DTranspose / SignalConv2D (+ GDN) stacks, not {type(transform).__name__}")
        self.layers = graph.layers
        last = self.layers[-1]
        if not isinstance(last, Conv) or last.act is not None:
            raise NotImplementedError("SGA backward expects a linear last layer")
        self.adj = []
        for i, l in enumerate(self.layers):
            if isinstance(l, GDN):
                if l.alpha != 1 or l.epsilon != 1.0:
                    raise NotImplementedError("SGA backward through GDN: alpha = 1, epsilon = 1 (GDN1 / the TFC 2.x default)")
                gamma, beta = transform._dev[f"{l.name}/gamma"], transform._dev[f"{l.name}/beta"]
                c = int(gamma.shape[0])
                g4 = gamma.reshape(1, 1, c, c).contiguous()
                self.adj.append((ops.ConvPlan("conv", g4, beta, 1, None, capi.PRO_ABS, capi.EPI_STORE),
                                 ops.ConvPlan("conv", g4, None, 1, None, capi.PRO_NONE, capi.EPI_STORE, kernel_io_swapped=True)))
                continue
            if l.kind not in ("convT", "sigup"):
                raise NotImplementedError(f"SGA backward through a {l.kind!r} layer")
            prev = self.layers[i - 1] if i > 0 else None
            prev_act = prev.act if isinstance(prev, Conv) else None
            if prev_act not in _MASKS:
                raise NotImplementedError(f"SGA backward through activation {prev_act!r}")
            epi = _MASKS[prev_act] if prev_act is not None else capi.EPI_STORE
            kernel = transform._dev[f"{l.name}/kernel"]
            if l.kind == "convT":    # kernel [kh,kw,Cout,Cin] of the transposed layer == HWIO kernel of its adjoint convolution
                self.adj.append(ops.ConvPlan("conv", kernel, None, l.s, None, capi.PRO_NONE, epi))
            else:                    # SignalConv2D up / down are adjoints of each other on the channel-swapped kernel array
                self.adj.append(ops.ConvPlan("sigdown", kernel, None, l.s, None, capi.PRO_NONE, epi, kernel_io_swapped=True))

    def forward(self, x):
        acts = []
        for l, a in zip(self.layers, self.adj):
            if isinstance(l, GDN):
                norm = a[0](x)
                acts.append((x, norm))
                x = ops.gdn_apply(x, norm, l.inverse)
            else:
                x = l(x)
                acts.append(x)
        return x, acts

    def backward(self, g, acts):
        for i in range(len(self.layers) - 1, -1, -1):
            l = self.layers[i]
            if isinstance(l, GDN):
                x, norm = acts[i]
                q, _ = ops.gdn_backward_prep(g, x, norm, l.inverse)
                g = ops.gdn_backward_finish(g, x, norm, self.adj[i][1](q), l.inverse)
                continue
            prev = self.layers[i - 1] if i > 0 else None
            masked = isinstance(prev, Conv) and prev.act is not None
            g = self.adj[i](g, res=acts[i - 1]) if masked else self.adj[i](g)
        return g


ConvTChain = GradChain      # former name


class TwoLayerBackward:
    """Forward-with-cache and input-gradient of TwoLayer[Res]Synthesis."""

    def __init__(self, t: _TwoLayerBase):
        if not t._merged:
            raise NotImplementedError("gradients through the two-layer synthesis exist for the shapes the reference's configs use "
                                      "(hidden width 12 / 24 / 48, 5x5 / 2 output layer, convolutional residual); this one runs forward only")
        self.t = t
        dev = t._w2.device
        w = t.get_weights()
        n1, nr, n2 = t._names
        k1 = w[f"{n1}/kernel"]
        if nr:
            k1 = np.concatenate([k1, w[f"{nr}/kernel"]], axis=2)
        c2 = k1.shape[2]
        self.cp = -(-c2 // 16) * 16                       # pad the gradient channels to a multiple of the 16-deep K stage (the vector
                                                          # gather path; 12 -> 16: half the adjoint's MFMAs of the 32 it used to pad to)
        k1p = np.zeros(k1.shape[:2] + (self.cp, k1.shape[3]), np.float32)
        k1p[:, :, :c2] = k1
        self.up_adj = ops.ConvPlan("conv", ops.to_device(k1p, dev), None, t._s[0])
        self.up_adj.algorithmic_cin = c2                  # the zero-padded gradient channels are not work (ConvPlan.flops)

    def forward(self, x):
        t = self.t
        mid = t._up(x)
        return ops.two_layer_tail(mid, t._ch, t._has_res, t._act_kind, t._beta, t._gamma, t._w2, t._b2, t._k[1], t._s[1]), mid

    def backward(self, g_xhat, mid):
        t = self.t
        g_h = ops.two_layer_out_adjoint(g_xhat, t._w2, t._ch, t._k[1], t._s[1])      # 3-channel stream, not a GEMM
        g_t = ops.two_layer_tail_bwd(mid, g_h, t._ch, t._has_res, t._act_kind, t._beta, t._gamma, self.cp)
        return self.up_adj(g_t)


def make_backward(transform):
    return TwoLayerBackward(transform) if isinstance(transform, _TwoLayerBase) else GradChain(transform)


class SGAEngine:
    """loss(z_loc, y_loc) = bpp + lambda * MSE_255 of frame_loss_given_latent_rvs(training=True) with the
    'sga' uq method, and its gradients w.r.t. (z_loc, y_loc) only (mshyper/models.py:397-399).  The factorized-prior model
    (factorized/models.py:108-118) has one latent: y~ = sga_round(y_loc), coded by the deep-factorized prior."""

    def __init__(self, model):
        self.m = model
        with torch.cuda.device(model.device):
            self.hyper = None if model.factorized else GradChain(model._hyper_synthesis)
            self.syn = make_backward(model._synthesis)

    def loss_and_grads(self, x, z_loc, y_loc, tau, rd_lambda, step=0, seed=0, noise_z=None, noise_y=None):
        m = self.m
        n, h, w, c = x.shape
        w_bpp = 1.0 / (n * h * w)                                      # bpp = mean_B(bits) / (H W)   (:302-307)
        scale = rd_lambda * 2.0 * 255.0 * 255.0 / (n * h * w * c)     # d(lambda * mean_B mean_HWC (255 d)^2)/d x_hat
        if m.factorized:
            y_t, sp_y, dby, bits_y = ops.sga_factorized_fwd(m._get_prior(), y_loc, tau, noise_y, seed, step)   # factorized :108-116
            recon, cache = self.syn.forward(y_t)
            g_x, sse = ops.distortion_grad(x, recon, scale)
            g_y = ops.sga_chain(self.syn.backward(g_x, cache), dby, sp_y, w_bpp)
            return dict(bits_z=torch.zeros_like(bits_y), bits_y=bits_y, sse=sse, g_z=None, g_y=g_y, recon=recon, z_tilde=None,
                        y_tilde=y_t)
        z_t, sp_z, dbz, bits_z = ops.sga_factorized_fwd(m._get_prior(), z_loc, tau, noise_z, seed, step)     # :262-268
        hyper, acts = self.hyper.forward(z_t)                                                              # :273
        y_t, sp_y, dv, dr, bits_y = ops.sga_normal_fwd(y_loc, hyper, tau, noise_y, seed, step)             # :285-291
        recon, cache = self.syn.forward(y_t)
        g_x, sse = ops.distortion_grad(x, recon, scale)                                                    # :313-317,343
        g_yt = self.syn.backward(g_x, cache)
        g_y, g_hyper = ops.sga_normal_bwd(g_yt, sp_y, dv, dr, w_bpp)
        g_zt = self.hyper.backward(g_hyper, acts)
        g_z = ops.sga_chain(g_zt, dbz, sp_z, w_bpp)
        return dict(bits_z=bits_z, bits_y=bits_y, sse=sse, g_z=g_z, g_y=g_y, recon=recon, z_tilde=z_t, y_tilde=y_t)
