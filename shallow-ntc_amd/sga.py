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
from .common._graph import Conv, Seq
from .common.transforms import _TwoLayerBase

_MASKS = {None: None, "relu": capi.EPI_MASK_RELU, "leaky_relu": capi.EPI_MASK_LEAKY}


class ConvTChain:
    """Forward-with-cache and input-gradient of a stack of Keras Conv2DTranspose layers
    (HyperSynthesis, JPEGLikeSynthesis, CNNSynthesis with relu / leaky-relu)."""

    def __init__(self, transform):
        graph = transform._graph
        if not isinstance(graph, Seq) or not all(isinstance(l, Conv) and l.kind == "convT" for l in graph.layers):
            raise NotImplementedError(f"SGA backward is implemented for Conv2DTranspose stacks, not {type(transform).__name__}")
        self.layers = graph.layers
        if self.layers[-1].act is not None:
            raise NotImplementedError("SGA backward expects a linear last layer")
        self.adj = []
        for i, l in enumerate(self.layers):
            prev_act = self.layers[i - 1].act if i > 0 else None
            if prev_act not in _MASKS:
                raise NotImplementedError(f"SGA backward through activation {prev_act!r}")
            epi = _MASKS[prev_act] if i > 0 and prev_act is not None else capi.EPI_STORE
            # kernel [kh,kw,Cout,Cin] of the transposed layer == HWIO kernel of its adjoint convolution
            self.adj.append(ops.ConvPlan("conv", transform._dev[f"{l.name}/kernel"], None, l.s, None, capi.PRO_NONE, epi))

    def forward(self, x):
        acts = []
        for l in self.layers:
            x = l(x)
            acts.append(x)
        return x, acts

    def backward(self, g, acts):
        for i in range(len(self.layers) - 1, -1, -1):
            g = self.adj[i](g, res=acts[i - 1]) if (i > 0 and self.layers[i - 1].act is not None) else self.adj[i](g)
        return g


class TwoLayerBackward:
    """Forward-with-cache and input-gradient of TwoLayer[Res]Synthesis."""

    def __init__(self, t: _TwoLayerBase):
        self.t = t
        dev = t._w2.device
        w = t.get_weights()
        n1, nr, n2 = t._names
        k1 = w[f"{n1}/kernel"]
        if nr:
            k1 = np.concatenate([k1, w[f"{nr}/kernel"]], axis=2)
        c2 = k1.shape[2]
        self.cp = -(-c2 // 32) * 32                       # pad the gradient channels to a 32 multiple (vector gather path)
        k1p = np.zeros(k1.shape[:2] + (self.cp, k1.shape[3]), np.float32)
        k1p[:, :, :c2] = k1
        self.up_adj = ops.ConvPlan("conv", ops.to_device(k1p, dev), None, t._s[0])

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
    return TwoLayerBackward(transform) if isinstance(transform, _TwoLayerBase) else ConvTChain(transform)


class SGAEngine:
    """loss(z_loc, y_loc) = bpp + lambda * MSE_255 of frame_loss_given_latent_rvs(training=True) with the
    'sga' uq method, and its gradients w.r.t. (z_loc, y_loc) only (mshyper/models.py:397-399)."""

    def __init__(self, model):
        self.m = model
        with torch.cuda.device(model.device):
            self.hyper = ConvTChain(model._hyper_synthesis)
            self.syn = make_backward(model._synthesis)

    def loss_and_grads(self, x, z_loc, y_loc, tau, rd_lambda, step=0, seed=0, noise_z=None, noise_y=None):
        m = self.m
        n, h, w, c = x.shape
        w_bpp = 1.0 / (n * h * w)                                      # bpp = mean_B(bits) / (H W)   (:302-307)
        scale = rd_lambda * 2.0 * 255.0 * 255.0 / (n * h * w * c)     # d(lambda * mean_B mean_HWC (255 d)^2)/d x_hat
        z_t, sp_z, dbz, bits_z = ops.sga_factorized_fwd(m._get_prior(), z_loc, tau, noise_z, seed, step)     # :262-268
        hyper, acts = self.hyper.forward(z_t)                                                              # :273
        y_t, sp_y, dv, dr, bits_y = ops.sga_normal_fwd(y_loc, hyper, tau, noise_y, seed, step)             # :285-291
        recon, cache = self.syn.forward(y_t) if isinstance(self.syn, TwoLayerBackward) else self._chain_fwd(y_t)
        g_x, sse = ops.distortion_grad(x, recon, scale)                                                    # :313-317,343
        g_yt = self.syn.backward(g_x, cache)
        g_y, g_hyper = ops.sga_normal_bwd(g_yt, sp_y, dv, dr, w_bpp)
        g_zt = self.hyper.backward(g_hyper, acts)
        g_z = ops.sga_chain(g_zt, dbz, sp_z, w_bpp)
        return dict(bits_z=bits_z, bits_y=bits_y, sse=sse, g_z=g_z, g_y=g_y, recon=recon, z_tilde=z_t, y_tilde=y_t)

    def _chain_fwd(self, y_t):
        out, acts = self.syn.forward(y_t)
        return out, acts
