"""Training step: ``Model.train_step`` of the reference (mshyper/models.py:375-383) =
tape.gradient of ``end_to_end_frame_loss(training=True)`` w.r.t. every trainable variable + Keras Adam with
``global_clipnorm`` and the CompressionSchedule learning rate (common/schedule.py:155-176).

There is no autograd here: the backward pass is written out layer by layer.
  * forward            the same gather-GEMM plans as inference (bias / activation / residual fused), activations kept;
  * input gradients    adjoint plans: the adjoint of Conv2D is Conv2DTranspose on the SAME kernel array and vice versa;
  * weight gradients   sntc_conv_wgrad (fp32 MFMA contraction over pixels), bias gradients sntc_bias_grad;
  * entropy terms      uniform-noise samples (uq method 'unoise'), analytic d bits / d (sample, mean, raw scale, prior);
  * optimizer          every variable lives in ONE flat float32 buffer (parameters, gradients, Adam moments), laid
                       out in backward order, so the data-parallel all-reduce (RCCL) runs on a few large contiguous
                       buckets that are launched as soon as their last gradient is written, overlapping the rest of
                       the backward pass; Adam is one kernel over the whole buffer.
Trainable here: Conv2D / Conv2DTranspose stacks, ELIC residual / attention blocks, the two-layer syntheses with
IGDN1 (reparameterised beta / gamma as in tfc.GDNParameter), the deep-factorized prior -- i.e. the reference's
two_layer_syn / two_layer_syn2 / jpegl training configs.  GDN-based analysis transforms and SignalConv2D (RDFT
kernels) are not.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from . import _capi as capi
from . import ops
from .common._graph import GDN, Conv, ResidualBlock, Seq, SimpleAttention
from .common.transforms import _TwoLayerBase

GDN_OFFSET = 2.0 ** -18           # tfc.GDNParameter: reparam_offset; pedestal = offset^2
GDN_BETA_MIN = 1e-6               # tfc.GDN beta_min; gamma's minimum is 0


class FlatStore:
    """All trainable variables in one flat device buffer (+ gradient and Adam moments of the same layout)."""

    def __init__(self, device):
        self.device = device
        self._arrays = OrderedDict()
        self.offsets = OrderedDict()
        self.marks = OrderedDict()         # bucket name -> end offset (backward order)

    def add(self, name, array):
        if name in self._arrays:
            raise KeyError(f"duplicate variable {name}")
        self._arrays[name] = np.ascontiguousarray(array, dtype=np.float32)

    def mark(self, bucket):
        self.marks[bucket] = len(self._arrays)       # turned into an end offset by finalize()

    def finalize(self):
        total, host, ends = 0, [], []
        for name, a in self._arrays.items():
            self.offsets[name] = (total, a.shape)
            total += a.size
            pad = (-total) % 4                       # keep every variable 16-byte aligned
            host.append(a.ravel())
            if pad:
                host.append(np.zeros(pad, np.float32))
                total += pad
            ends.append(total)
        self.total = total
        self.marks = OrderedDict((b, ends[cnt - 1] if cnt else 0) for b, cnt in self.marks.items())
        flat = np.concatenate(host) if host else np.zeros(0, np.float32)
        with torch.cuda.device(self.device):
            self.param = ops.to_device(flat, self.device)
            self.grad = torch.zeros_like(self.param)
            self.m = torch.zeros_like(self.param)
            self.v = torch.zeros_like(self.param)
        self._arrays = None
        return self

    def _view(self, buf, name):
        off, shape = self.offsets[name]
        return buf[off:off + int(np.prod(shape))].view(shape)

    def p(self, name):
        return self._view(self.param, name)

    def g(self, name):
        return self._view(self.grad, name)

    def export(self, buf=None):
        host = (self.param if buf is None else buf).cpu().numpy()
        return OrderedDict((k, host[o:o + int(np.prod(s))].reshape(s).copy()) for k, (o, s) in self.offsets.items())


class TConv:
    """One trainable convolution: forward plan, adjoint plan, weight / bias gradients.

    Keras Conv2D / Conv2DTranspose train their kernel directly.  tfc.SignalConv2D ("sigdown" / "sigup",
    common/transforms.py:101-175) trains the real-DFT coefficients of its kernel (tfc.RDFTParameter, the layer's default):
    the store holds ``<name>/rdft`` [rows, cin * cout]; kernel = M rdft is recomputed at every refresh and the weight
    gradient is pulled back with M^T (M = tf_checkpoint.irdft_matrix of the kernel's spatial shape), so Adam moves the
    same variables as in the reference."""

    def __init__(self, store, name, kind, k, s, cin, cout, act, bias, add_res=False, adj_epilogue=capi.EPI_STORE, rdft_basis=None):
        self.name, self.kind, self.k, self.s, self.cin, self.cout, self.act = name, kind, k, s, cin, cout, act
        self.signal = kind in ("sigdown", "sigup")
        if self.signal:
            if k % 2 == 0:
                raise NotImplementedError(f"{name}: even SignalConv2D kernels are not used by the reference")
            self.M = rdft_basis                                        # [k * k, rows] device tensor
            self.rdft, self.g_rdft = store.p(f"{name}/rdft"), store.g(f"{name}/rdft")
            self.W = torch.empty((k, k, cin, cout), dtype=torch.float32, device=self.rdft.device)
            self.gW = torch.empty_like(self.W)
            self._gT = torch.empty((k, k, cout, cin), dtype=torch.float32, device=self.rdft.device) if kind == "sigup" else None
            ops.small_matmul(self.M, self.rdft, self.W.view(k * k, cin * cout))
        else:
            self.W, self.gW = store.p(f"{name}/kernel"), store.g(f"{name}/kernel")
        self.b = store.p(f"{name}/bias") if bias else None
        self.gb = store.g(f"{name}/bias") if bias else None
        self.fwd_plan = ops.ConvPlan(kind, self.W, self.b, s, act, capi.PRO_NONE, capi.EPI_ADD if add_res else capi.EPI_STORE)
        # kernel [kh,kw,Cin,Cout] of a Conv2D == kernel [kh,kw,Cout',Cin'] of its adjoint Conv2DTranspose (and vice versa);
        # SignalConv2D down / up are adjoints of each other on the channel-swapped kernel array
        # adj_epilogue: what the input-gradient launch does on the way out --
        #   EPI_MASK_RELU  multiply by 1[x > 0], x = this layer's input: the producer's relu backward rides along
        #   EPI_ADD        add a second gradient (the skip path of a ResidualBlock)
        self.adj_epilogue = adj_epilogue
        self.adj_kind = {"conv": "convT", "convT": "conv", "sigdown": "sigup", "sigup": "sigdown"}[kind]
        self.adj_plan = ops.ConvPlan(self.adj_kind, self.W, None, s, None, capi.PRO_NONE, adj_epilogue, kernel_io_swapped=self.signal)

    def prepare(self):
        """What must be recomputed from the variables before the plans are re-packed (tfc's RDFT-stored kernels)."""
        if self.signal:
            ops.small_matmul(self.M, self.rdft, self.W.view(self.k * self.k, self.cin * self.cout))

    def plan_entries(self):
        """(plan, weight array, bias array) of every plan that follows this layer's variables (ops.PlanGroup)."""
        return [(self.fwd_plan, self.W, self.b), (self.adj_plan, self.W, None)]

    def refresh(self):
        self.prepare()
        self.fwd_plan.update(self.W, self.b)
        self.adj_plan.update(self.W, None)

    def fwd(self, x, res=None):
        y = self.fwd_plan(x, res)
        return y, (x, y)

    def bwd(self, ctx, g, need_dx=True, act_folded=False, adj_res=None):
        x, y = ctx
        if self.act is not None and not act_folded:
            g = ops.act_backward(g, y, self.act)
        side = ops.WGRAD_STREAM
        if side is None:
            self._param_grads(x, g)
        else:
            # d loss / d kernel and d loss / d bias depend on (x, g) only; the input gradient below depends on g only: the
            # two run side by side (on the small maps of the deep layers neither fills the device alone).  The trainer
            # joins the side stream before a bucket's gradients are used.
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                self._param_grads(x, g)
            x.record_stream(side)              # the caching allocator must not hand these to main-stream work early
            g.record_stream(side)
        if not need_dx:
            return None
        if self.adj_epilogue == capi.EPI_MASK_RELU:
            return self.adj_plan(g, res=x)
        if self.adj_epilogue == capi.EPI_ADD:
            return self.adj_plan(g, res=adj_res)
        return self.adj_plan(g)

    def _param_grads(self, x, g):
        if self.kind == "sigup":               # comes out [k, k, Cout, Cin]: transpose to the layer's [k, k, Cin, Cout]
            ops.conv_wgrad(self.kind, self.k, self.s, self.cin, self.cout, x, g, self._gT)
            ops.transpose_last2(self._gT.view(self.k * self.k, self.cout, self.cin), self.gW)
        else:
            ops.conv_wgrad(self.kind, self.k, self.s, self.cin, self.cout, x, g, self.gW)
        if self.signal:                         # d loss / d rdft = M^T d loss / d kernel
            ops.small_matmul(self.M, self.gW.view(self.k * self.k, self.cin * self.cout), self.g_rdft, transpose_a=True)
        if self.gb is not None:
            ops.bias_grad(g, self.gb)

    def convs(self):
        return [self]


class TSeq:
    def __init__(self, items):
        self.items = items
        # conv (relu) -> conv: the consumer's input-gradient launch applies the producer's relu mask
        self.folded = [False] * len(items)
        for i in range(len(items) - 1):
            a, b = items[i], items[i + 1]
            if isinstance(a, TConv) and isinstance(b, TConv) and a.act == "relu" and b.adj_epilogue == capi.EPI_STORE:
                b.adj_epilogue = capi.EPI_MASK_RELU
                b.adj_plan = ops.ConvPlan(b.adj_kind, b.W, None, b.s, None, capi.PRO_NONE, capi.EPI_MASK_RELU, kernel_io_swapped=b.signal)
                self.folded[i] = True

    def fwd(self, x):
        ctxs = []
        for it in self.items:
            x, c = it.fwd(x)
            ctxs.append(c)
        return x, ctxs

    def bwd(self, ctxs, g, need_dx=True):
        for i in range(len(self.items) - 1, -1, -1):
            if self.folded[i]:
                g = self.items[i].bwd(ctxs[i], g, need_dx or i > 0, act_folded=True)
            else:
                g = self.items[i].bwd(ctxs[i], g, need_dx or i > 0)
        return g

    def convs(self):
        return [c for it in self.items for c in it.convs()]


class TGDN:
    """tfc.GDN / the reference's GDN1 inside an analysis / synthesis stack (common/transforms.py:8-63,150,170), alpha = 1,
    epsilon = 1:  y = x / (beta + |x| gamma)  (inverse: x * (...)), with tfc's non-negative reparameterisation of beta
    and gamma.  The norm pool and its adjoint are 1x1 gather-GEMM plans on the effective gamma; d gamma = |x|^T q is a
    1x1 weight gradient, d beta a column sum (q = d loss / d norm)."""

    def __init__(self, store, name, channels, inverse):
        self.name, self.c, self.inverse = name, channels, bool(inverse)
        self.beta_raw, self.g_beta_raw = store.p(f"{name}/beta_raw"), store.g(f"{name}/beta_raw")
        self.gamma_raw, self.g_gamma_raw = store.p(f"{name}/gamma_raw"), store.g(f"{name}/gamma_raw")
        self.beta, self.gamma = torch.empty_like(self.beta_raw), torch.empty_like(self.gamma_raw)
        self.g_beta, self.g_gamma = torch.empty_like(self.beta_raw), torch.empty_like(self.gamma_raw)
        self.pedestal = GDN_OFFSET ** 2
        self.beta_bound = math.sqrt(GDN_BETA_MIN + self.pedestal)
        self.gamma_bound = GDN_OFFSET
        self._effective()
        g4 = self.gamma.view(1, 1, channels, channels)                   # gamma[in, out] = a 1x1 kernel [1, 1, Cin, Cout]
        self.norm_plan = ops.ConvPlan("conv", g4, self.beta, 1, None, capi.PRO_ABS, capi.EPI_STORE)
        self.adj_plan = ops.ConvPlan("conv", g4, None, 1, None, capi.PRO_NONE, capi.EPI_STORE, kernel_io_swapped=True)

    def _effective(self):
        capi.call("sntc_gdn_reparam_forward", ops._ptr(self.beta_raw), self.beta_raw.numel(), self.beta_bound, self.pedestal,
                  ops._ptr(self.beta), ops._stream())
        capi.call("sntc_gdn_reparam_forward", ops._ptr(self.gamma_raw), self.gamma_raw.numel(), self.gamma_bound, self.pedestal,
                  ops._ptr(self.gamma), ops._stream())

    def prepare(self):
        self._effective()

    def plan_entries(self):
        g4 = self.gamma.view(1, 1, self.c, self.c)
        return [(self.norm_plan, g4, self.beta), (self.adj_plan, g4, None)]

    def refresh(self):
        self._effective()
        g4 = self.gamma.view(1, 1, self.c, self.c)
        self.norm_plan.update(g4, self.beta)
        self.adj_plan.update(g4, None)

    def fwd(self, x):
        norm = self.norm_plan(x)
        return ops.gdn_apply(x, norm, self.inverse), (x, norm)

    def bwd(self, ctx, g, need_dx=True):
        x, norm = ctx
        q, ax = ops.gdn_backward_prep(g, x, norm, self.inverse)
        ops.conv_wgrad("conv", 1, 1, self.c, self.c, ax, q, self.g_gamma)
        ops.bias_grad(q, self.g_beta)
        capi.call("sntc_gdn_reparam_backward", ops._ptr(self.beta_raw), ops._ptr(self.g_beta), self.g_beta.numel(), self.beta_bound,
                  ops._ptr(self.g_beta_raw), ops._stream())
        capi.call("sntc_gdn_reparam_backward", ops._ptr(self.gamma_raw), ops._ptr(self.g_gamma), self.g_gamma.numel(),
                  self.gamma_bound, ops._ptr(self.g_gamma_raw), ops._stream())
        if not need_dx:
            return None
        return ops.gdn_backward_finish(g, x, norm, self.adj_plan(q), self.inverse)

    def convs(self):
        return [self]            # takes part in the refresh sweep


class TResidualBlock:
    """x + conv1x1(relu(conv3x3(relu(conv1x1(x)))))  (elic.py:41-68); the skip rides on the last conv's epilogue."""

    def __init__(self, c0, c1, c2):
        self.c = (c0, c1, c2)

    def fwd(self, x):
        h0, k0 = self.c[0].fwd(x)
        h1, k1 = self.c[1].fwd(h0)
        y, k2 = self.c[2].fwd(h1, res=x)
        return y, (k0, k1, k2)

    def bwd(self, ctx, g, need_dx=True):
        # the relu backward of conv0 / conv1 rides on the input-gradient launch of the layer above (mask epilogue),
        # the skip gradient on conv0's (add epilogue): 3 weight gradients + 3 input gradients + 3 bias sums, nothing else
        d = self.c[2].bwd(ctx[2], g)                               # -> already masked by conv1's relu
        d = self.c[1].bwd(ctx[1], d, act_folded=True)              # -> already masked by conv0's relu
        return self.c[0].bwd(ctx[0], d, act_folded=True, adj_res=g)

    def convs(self):
        return list(self.c)


class TAttention:
    """x + trunk(x) * sigmoid(conv1x1(branch(x)))  (elic.py:71-100), gate unfused so that the sigmoid output is kept."""

    def __init__(self, trunk, branch, gate):
        self.trunk, self.branch, self.gate = TSeq(trunk), TSeq(branch), gate

    def fwd(self, x):
        t, kt = self.trunk.fwd(x)
        b, kb = self.branch.fwd(x)
        s, kg = self.gate.fwd(b)
        return ops.gate_forward(x, t, s), (kt, kb, kg, t, s)

    def bwd(self, ctx, g, need_dx=True):
        kt, kb, kg, t, s = ctx
        g_t, g_spre = ops.gate_backward(g, t, s)
        d_b = self.gate.bwd(kg, g_spre, act_folded=True)
        dx = self.trunk.bwd(kt, g_t)
        ops.axpy(dx, self.branch.bwd(kb, d_b))
        return ops.axpy(dx, g)

    def convs(self):
        return self.trunk.convs() + self.branch.convs() + [self.gate]


class TTwoLayer:
    """TwoLayer[Res]Synthesis (transforms.py:298-361) unfused for training: one stride-8 transposed conv producing
    [base | res], the hidden layer (IGDN1 + add), the 5x5/2 output layer."""

    def __init__(self, store, pre, t: _TwoLayerBase, cin):
        if not t._merged:
            raise NotImplementedError("gradients through the two-layer synthesis exist for the shapes the reference's configs use "
                                      "(hidden width 12 / 24 / 48, 5x5 / 2 output layer, convolutional residual); this one runs forward only")
        self.t, self.pre = t, pre
        c2 = t._ch * (2 if t._has_res else 1)
        self.up = TConv(store, f"{pre}/up", "convT", t._k[0], t._s[0], cin, c2, None, True)
        self.out = TConv(store, f"{pre}/{t._names[2]}", "convT", t._k[1], t._s[1], t._ch, t._out_ch, None, True)
        self.gdn = t._act_kind == 1
        if t._act_kind == 2:
            raise NotImplementedError("a forward GDN1 hidden activation is not trainable here (the reference configs use igdn)")
        if self.gdn:
            self.beta_raw, self.g_beta_raw = store.p(f"{pre}/act/beta_raw"), store.g(f"{pre}/act/beta_raw")
            self.gamma_raw, self.g_gamma_raw = store.p(f"{pre}/act/gamma_raw"), store.g(f"{pre}/act/gamma_raw")
            self.beta, self.gamma = torch.empty_like(self.beta_raw), torch.empty_like(self.gamma_raw)
            self.g_beta, self.g_gamma = torch.empty_like(self.beta_raw), torch.empty_like(self.gamma_raw)
            self.pedestal = GDN_OFFSET ** 2
            self.beta_bound = math.sqrt(GDN_BETA_MIN + self.pedestal)
            self.gamma_bound = GDN_OFFSET
        else:
            self.beta = self.gamma = None

    def refresh(self):
        if self.gdn:
            capi.call("sntc_gdn_reparam_forward", ops._ptr(self.beta_raw), self.beta_raw.numel(), self.beta_bound, self.pedestal,
                      ops._ptr(self.beta), ops._stream())
            capi.call("sntc_gdn_reparam_forward", ops._ptr(self.gamma_raw), self.gamma_raw.numel(), self.gamma_bound, self.pedestal,
                      ops._ptr(self.gamma), ops._stream())

    def fwd(self, x):
        t = self.t
        mid, k_up = self.up.fwd(x)
        h = ops.two_layer_hidden(mid, t._ch, t._has_res, t._act_kind, self.beta, self.gamma)
        out, k_out = self.out.fwd(h)
        return out, (k_up, k_out, mid)

    def bwd(self, ctx, g, need_dx=True):
        t = self.t
        k_up, k_out, mid = ctx
        self.out.bwd(k_out, g, need_dx=False)                      # weight / bias gradients of the output layer
        g_h = ops.two_layer_out_adjoint(g, self.out.W, t._ch, t._k[1], t._s[1])
        c2 = t._ch * (2 if t._has_res else 1)
        if self.gdn:
            g_t, ax, gx = ops.two_layer_tail_bwd(mid, g_h, t._ch, t._has_res, t._act_kind, self.beta, self.gamma, c2, param_operands=True)
            # d gamma[i, j] = sum_p |x_i| (g x)_j: a 1x1 "convolution" weight gradient; d beta = column sums of g x
            ops.conv_wgrad("conv", 1, 1, t._ch, t._ch, ax, gx, self.g_gamma)
            ops.bias_grad(gx, self.g_beta)
            capi.call("sntc_gdn_reparam_backward", ops._ptr(self.beta_raw), ops._ptr(self.g_beta), self.g_beta.numel(), self.beta_bound,
                      ops._ptr(self.g_beta_raw), ops._stream())
            capi.call("sntc_gdn_reparam_backward", ops._ptr(self.gamma_raw), ops._ptr(self.g_gamma), self.g_gamma.numel(),
                      self.gamma_bound, ops._ptr(self.g_gamma_raw), ops._stream())
        else:
            g_t = ops.two_layer_tail_bwd(mid, g_h, t._ch, t._has_res, t._act_kind, None, None, c2)
        return self.up.bwd(k_up, g_t, need_dx)

    def convs(self):
        return [self.up, self.out]


def gdn_raw(effective, minimum):
    pedestal = GDN_OFFSET ** 2
    return np.sqrt(np.maximum(np.asarray(effective, np.float64) + pedestal, pedestal)).astype(np.float32)


def gdn_effective(raw, minimum):
    pedestal = GDN_OFFSET ** 2
    return (np.maximum(np.asarray(raw, np.float64), math.sqrt(minimum + pedestal)) ** 2 - pedestal).astype(np.float32)


def checkpoints_to_keep(age, current, max_to_keep):
    """tf.train.CheckpointManager(max_to_keep=N) retention (reference common/train_lib.py:124-126): the checkpoint just written
    (step ``current``) always stays, next to the N - 1 most RECENT others.  ``age``: {step: modification time} of the bundles in
    the directory (``current`` may or may not be among them).  Returns the surviving steps, oldest first, ``current`` last."""
    others = sorted((k for k in age if k != current), key=lambda k: (age[k], k))
    nkeep = max(1, int(max_to_keep)) - 1                       # survivors besides the checkpoint just written
    keep = others[max(0, len(others) - nkeep):] if nkeep > 0 else []      # (a negative slice start would count from the end)
    return keep + [current]


class Trainer:
    """Owns the training state of a mean-scale hyperprior ``Model`` and runs ``train_step``."""

    BUCKETS = ("synthesis", "prior", "hyper_synthesis", "hyper_analysis", "analysis")   # backward order

    def __init__(self, model, seed=0):
        self.factorized = bool(model.factorized)
        if self.factorized:                 # factorized/models.py: no hyper transforms, the prior codes y itself
            self.BUCKETS = ("synthesis", "prior", "analysis")
        uq = model._latent_config["uq"].get("method", "unoise")
        if self.factorized and uq != "unoise":
            raise NotImplementedError("the factorized-prior model trains with uniform noise (factorized/configs/bls2017.py)")
        if uq not in ("unoise", "mixedq"):
            raise NotImplementedError(f"training with uq method {uq!r} (the shipped training configs use 'unoise' and 'mixedq')")
        self.uq = uq
        self.m = model
        self.device = model.device
        self.seed = seed
        self.use_graph = True
        self.overlap_wgrad = True        # weight / bias gradients on a side stream, next to the input-gradient chain
        self.store = FlatStore(self.device)
        w = model.get_weights()
        tr = model._transforms()
        b, hb = model._bottleneck_size, model._hyper_bottleneck_size
        self._two_layer = isinstance(tr["synthesis"], _TwoLayerBase)
        self._rdft = {}                      # kernel size -> real-DFT basis of tfc.RDFTParameter on the device
        self._reparam = OrderedDict()        # store name -> (weights name, kind) for variables that are not stored as is
        # ---- variable inventory, in backward order (see module docstring) ----
        for bucket in self.BUCKETS:
            if bucket == "prior":
                nl = len(model._prior_num_filters) + 1
                for kind, cnt in (("matrix", nl), ("bias", nl), ("factor", nl - 1)):      # per kind, layer order: the C-ABI layout
                    for k in range(cnt):
                        self.store.add(f"prior/{kind}_{k}", w[f"prior/{kind}_{k}"])
            elif bucket == "synthesis" and self._two_layer:
                self._add_two_layer(tr["synthesis"], w)
            else:
                special = self._special_variables(tr[bucket]._graph, bucket)
                names = [k for k in w if k.startswith(bucket + "/")]
                for k in reversed(names):
                    if k in special:                       # SignalConv2D kernel -> rdft, GDN beta / gamma -> raw variable
                        kind = special[k]
                        if kind == "rdft":
                            from .common.tf_checkpoint import kernel_to_rdft
                            sk, val = k[:-len("kernel")] + "rdft", kernel_to_rdft(w[k])
                        else:
                            sk, val = k + "_raw", gdn_raw(w[k], GDN_BETA_MIN if kind == "beta" else 0.0)
                        self._reparam[sk] = (k, kind)
                        self.store.add(sk, val)
                    else:
                        self.store.add(k, w[k])
            self.store.mark(bucket)
        self.store.finalize()
        with torch.cuda.device(self.device):
            self.analysis = self._build(tr["analysis"]._graph, "analysis", 3)[0]
            if not self.factorized:
                self.hyper_analysis = self._build(tr["hyper_analysis"]._graph, "hyper_analysis", b)[0]
                self.hyper_synthesis = self._build(tr["hyper_synthesis"]._graph, "hyper_synthesis", hb)[0]
            if self._two_layer:
                self.synthesis = TTwoLayer(self.store, "synthesis", tr["synthesis"], b)
            else:
                self.synthesis = self._build(tr["synthesis"]._graph, "synthesis", b)[0]
            self._prior = model._get_prior()
            if model._prior_channels() % 4:
                raise NotImplementedError("the flat prior layout needs a hyper-latent channel count divisible by 4")
            self._prior_names = dict(m=self.store.offsets["prior/matrix_0"][0], b=self.store.offsets["prior/bias_0"][0],
                                     f=self.store.offsets.get("prior/factor_0", (0, ()))[0])
            self._grad_rec = torch.empty((capi.load().sntc_prior_record_floats(self._prior._h),), dtype=torch.float32, device=self.device)
            self._refresh()
        self.step_count = int(model._step)      # a restored / warm-started model continues its schedules and Adam bias correction
        self._handles = []

    # ---- construction -----------------------------------------------------------------------------
    def _add_two_layer(self, t, w):
        n1, nr, n2 = t._names
        k1, b1 = w[f"synthesis/{n1}/kernel"], w[f"synthesis/{n1}/bias"]
        if nr:
            k1 = np.concatenate([k1, w[f"synthesis/{nr}/kernel"]], axis=2)
            b1 = np.concatenate([b1, w[f"synthesis/{nr}/bias"]])
        self.store.add(f"synthesis/{n2}/kernel", w[f"synthesis/{n2}/kernel"])
        self.store.add(f"synthesis/{n2}/bias", w[f"synthesis/{n2}/bias"])
        if t._act_kind == 1:
            self.store.add("synthesis/act/beta_raw", gdn_raw(w["synthesis/act/beta"], GDN_BETA_MIN))
            self.store.add("synthesis/act/gamma_raw", gdn_raw(w["synthesis/act/gamma"], 0.0))
        self.store.add("synthesis/up/kernel", k1)
        self.store.add("synthesis/up/bias", b1)

    @staticmethod
    def _special_variables(graph, bucket):
        """{weights name: "rdft" | "beta" | "gamma"} of a transform: the variables tfc trains through a reparameterisation."""
        out = {}

        def walk(node):
            if isinstance(node, Conv) and node.kind in ("sigdown", "sigup"):
                out[f"{bucket}/{node.name}/kernel"] = "rdft"
            elif isinstance(node, GDN):
                out[f"{bucket}/{node.name}/beta"] = "beta"
                out[f"{bucket}/{node.name}/gamma"] = "gamma"
            elif isinstance(node, Seq):
                for l in node.layers:
                    walk(l)

        walk(graph)
        return out

    def _basis(self, k):
        if k not in self._rdft:
            from .common.tf_checkpoint import irdft_matrix
            self._rdft[k] = ops.to_device(irdft_matrix((k, k)), self.device)
        return self._rdft[k]

    def _conv(self, pre, node, cin, add_res=False, adj=capi.EPI_STORE):
        basis = self._basis(node.k) if node.kind in ("sigdown", "sigup") else None
        return TConv(self.store, f"{pre}/{node.name}", node.kind, node.k, node.s, cin, node.cout, node.act, node.bias, add_res, adj,
                     rdft_basis=basis)

    def _build(self, node, pre, cin):
        if isinstance(node, Conv):
            return self._conv(pre, node, cin), node.cout
        if isinstance(node, Seq):
            items = []
            for l in node.layers:
                it, cin = self._build(l, pre, cin)
                items.append(it)
            return TSeq(items), cin
        if isinstance(node, ResidualBlock):
            a, bb, c = node._mk(cin)
            return TResidualBlock(self._conv(pre, a, cin, adj=capi.EPI_ADD), self._conv(pre, bb, cin // 2, adj=capi.EPI_MASK_RELU),
                                  self._conv(pre, c, cin // 2, add_res=True, adj=capi.EPI_MASK_RELU)), cin
        if isinstance(node, SimpleAttention):
            trunk, branch, gate = node._mk(cin)
            g = Conv(gate.name, "conv", cin, 1, 1, "sigmoid")               # plain epilogue: the gate is applied by gate_forward
            return TAttention([self._build(r, pre, cin)[0] for r in trunk], [self._build(r, pre, cin)[0] for r in branch],
                              self._conv(pre, g, cin)), cin
        if isinstance(node, GDN):
            if node.alpha != 1 or node.epsilon != 1.0:
                raise NotImplementedError("training implements tfc.GDN with alpha = 1, epsilon = 1 (the reference's GDN1 and the "
                                          "TFC 2.x constructor default); the classic alpha = 2, epsilon = 0.5 form is inference only")
            return TGDN(self.store, f"{pre}/{node.name}", cin, node.inverse), cin
        raise NotImplementedError(type(node).__name__)

    def _all_convs(self):
        hyper = [] if self.factorized else self.hyper_analysis.convs() + self.hyper_synthesis.convs()
        return self.analysis.convs() + hyper + self.synthesis.convs()

    def _refresh(self):
        """Parameters changed: re-pack every plan, recompute the effective GDN parameters and the prior record.
        ~170 plans x (pack kernels + bias copy) is ~400 tiny launches: after the first (eager) pass the sequence is
        replayed as one captured HIP graph -- all pointers are views of the flat store, nothing allocates."""
        graph = getattr(self, "_refresh_graph", None)
        if graph is not None:
            graph.replay()
            return
        self._refresh_eager()
        if getattr(self, "_refresh_warm", 0) >= 1 and self.use_graph:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._refresh_eager()
            self._refresh_graph = g
        self._refresh_warm = getattr(self, "_refresh_warm", 0) + 1

    def _refresh_eager(self):
        """Every plan follows its variables: the per-layer preparations (RDFT kernels, GDN reparameterisation), then ONE
        re-pack launch for all plans (ops.PlanGroup) instead of ~400 pack / copy launches."""
        layers = self._all_convs()
        for c in layers:
            c.prepare()
        if getattr(self, "_plan_group", None) is None:
            self._plan_group = ops.PlanGroup([e for c in layers for e in c.plan_entries()])
        self._plan_group.update()
        if self._two_layer:
            self.synthesis.refresh()
        p = self.store.param
        o = self._prior_names
        capi.call("sntc_prior_update", self._prior._h, ops._ptr(p[o["m"]:]), ops._ptr(p[o["b"]:]), ops._ptr(p[o["f"]:]), ops._stream())

    # ---- one step -----------------------------------------------------------------------------------
    def loss_and_grads(self, x, rd_lambda, noise_z=None, noise_y=None, on_bucket=None):
        """Forward + backward; leaves d loss / d variable in ``store.grad`` and returns the loss terms.
        ``on_bucket(name)`` is called as soon as all gradients of a bucket are written (backward order)."""
        m = self.m
        n, h, w, c = x.shape
        f = m.downsample_factor
        if h % f or w % f:
            raise ValueError(f"training patches must be multiples of {f} (the reference trains on 256 x 256 crops)")
        on_bucket = on_bucket or (lambda name: None)
        side = None
        if self.overlap_wgrad:
            if getattr(self, "_wgrad_stream", None) is None:
                self._wgrad_stream = ops.side_streams(1, x.device)[0]
            side = self._wgrad_stream

        def notify(name):
            if side is not None:               # the bucket's weight gradients were written on the side stream
                torch.cuda.current_stream().wait_stream(side)
            on_bucket(name)

        ops.WGRAD_STREAM = side
        try:
            return self._loss_and_grads(x, rd_lambda, noise_z, noise_y, notify)
        finally:
            ops.WGRAD_STREAM = None

    def _loss_and_grads(self, x, rd_lambda, noise_z, noise_y, notify):
        n, h, w, c = x.shape
        w_bpp = 1.0 / (n * h * w)                                      # bpp = mean_B(bits) / (H W)        (:302-307)
        scale = rd_lambda * 2.0 * 255.0 * 255.0 / (n * h * w * c)     # d(lambda mean (255 d)^2) / d x_hat  (:313-317,343)
        step = self.step_count
        y, k_a = self.analysis.fwd(x)                                                              # :218-222
        if self.factorized:
            # factorized/models.py:101-118 with training=True: y~ = y + U(-.5, .5) is coded by the deep-factorized prior
            # and decoded by the synthesis; no hyper transforms
            y_t = ops.noise_add(y, noise_y, self.seed, 2 * step + 1)
            dby = torch.empty_like(y_t)
            bits_y = torch.empty((n,), dtype=torch.float64, device=x.device)
            capi.call("sntc_noisy_factorized", self._prior._h, ops._ptr(y_t), n, y_t.shape[1] * y_t.shape[2], ops._ptr(dby),
                      ops._ptr(self._grad_rec), ops._ptr(bits_y), ops._stream())
            recon, k_s = self.synthesis.fwd(y_t)
            g_x, sse = ops.distortion_grad(x, recon, scale)
            g_y = self.synthesis.bwd(k_s, g_x)
            notify("synthesis")
            st, o = self.store, self._prior_names
            capi.call("sntc_prior_param_grad", self._prior._h, ops._ptr(st.param[o["m"]:]), ops._ptr(st.param[o["f"]:]),
                      ops._ptr(self._grad_rec), w_bpp, ops._ptr(st.grad[o["m"]:]), ops._ptr(st.grad[o["b"]:]),
                      ops._ptr(st.grad[o["f"]:]), ops._stream())
            notify("prior")
            ops.axpy(g_y, dby, w_bpp)
            self.analysis.bwd(k_a, g_y, need_dx=False)
            notify("analysis")
            return dict(bits_z=torch.zeros_like(bits_y), bits_y=bits_y, sse=sse, recon=recon, y=y, z=None)
        z, k_ha = self.hyper_analysis.fwd(y)
        z_t = ops.noise_add(z, noise_z, self.seed, 2 * step)                                       # :253-256 (training=True)
        dbz = torch.empty_like(z_t)
        bits_z = torch.empty((n,), dtype=torch.float64, device=x.device)
        capi.call("sntc_noisy_factorized", self._prior._h, ops._ptr(z_t), n, z_t.shape[1] * z_t.shape[2], ops._ptr(dbz),
                  ops._ptr(self._grad_rec), ops._ptr(bits_z), ops._stream())
        # 'unoise': the noisy sample also feeds the decoder side; 'mixedq' (:257-259,281-283): rates at the noisy values,
        # decoder side sees the hard-rounded values through a straight-through estimator -- the backward pass is the
        # same in both cases (d sample / d loc = 1, d sample / d mu = 0)
        mixed = self.uq == "mixedq"
        if mixed:
            from .entropy_coding import int_to_float, round_to_int
            z_dec = int_to_float(round_to_int(z))
        else:
            z_dec = z_t
        hyper, k_hs = self.hyper_synthesis.fwd(z_dec)                                              # :273
        y_t = ops.noise_add(y, noise_y, self.seed, 2 * step + 1)                                   # :277-280
        bits_y, dv, dr = ops.noisy_normal(y_t, hyper)
        y_dec = ops.entropy_scale_normal(y, hyper, False)[0] if mixed else y_t                     # round(y - mu) + mu
        recon, k_s = self.synthesis.fwd(y_dec)                                                     # :297
        g_x, sse = ops.distortion_grad(x, recon, scale)
        # ---- backward ----
        g_yt = self.synthesis.bwd(k_s, g_x)
        notify("synthesis")
        st, o = self.store, self._prior_names
        capi.call("sntc_prior_param_grad", self._prior._h, ops._ptr(st.param[o["m"]:]), ops._ptr(st.param[o["f"]:]),
                  ops._ptr(self._grad_rec), w_bpp, ops._ptr(st.grad[o["m"]:]), ops._ptr(st.grad[o["b"]:]), ops._ptr(st.grad[o["f"]:]),
                  ops._stream())
        notify("prior")
        g_y, g_hyper = ops.sga_normal_bwd(g_yt, None, dv, dr, w_bpp)
        g_z = self.hyper_synthesis.bwd(k_hs, g_hyper)
        notify("hyper_synthesis")
        ops.axpy(g_z, dbz, w_bpp)
        ops.axpy(g_y, self.hyper_analysis.bwd(k_ha, g_z))
        notify("hyper_analysis")
        self.analysis.bwd(k_a, g_y, need_dx=False)
        notify("analysis")
        return dict(bits_z=bits_z, bits_y=bits_y, sse=sse, recon=recon, y=y, z=z)

    def _bucket_slices(self):
        out, lo = OrderedDict(), 0
        for name, hi in self.store.marks.items():
            out[name] = (lo, hi)
            lo = hi
        return out

    def train_step(self, x, noise_z=None, noise_y=None):
        """One optimizer step on this rank's batch ``x`` (NHWC float32 in [-0.5, 0.5], device tensor or ndarray).
        Under torch.distributed every rank passes its own slice of the global batch; gradients are averaged."""
        from .distributed import BucketReducer
        m = self.m
        x = m._as_device_images(x)
        with torch.cuda.device(self.device):
            rd_lambda = m._scheduled_rd_lambda
            lr = m._scheduled_lr
            reducer = BucketReducer(self.store.grad, self._bucket_slices())
            out = self.loss_and_grads(x, rd_lambda, noise_z, noise_y, on_bucket=reducer.launch)
            inv_world = reducer.finish()
            clip = m._optimizer_config.get("global_clipnorm")
            # ONE device -> host copy of everything the step's control flow needs, BEFORE any state is touched
            host = torch.stack([out["bits_z"], out["bits_y"], out["sse"]]).cpu().numpy()
            ops.check_conv_status()              # a flagged stream-K launch raises before the optimizer touches anything
            norm = math.sqrt(float(ops.sumsq(self.store.grad).item())) * inv_world
            n, h, w, c = x.shape
            bpp = float(host[0].mean() / (h * w) + host[1].mean() / (h * w))
            mse_i = host[2] / (h * w * c)
            mse = float(mse_i.mean())
            loss = bpp + rd_lambda * mse
            if not (math.isfinite(loss) and math.isfinite(norm)):
                # tf.debugging.check_numerics inside the loss (mshyper/models.py:308-309,356) fires before apply_gradients:
                # parameters, Adam moments and the packed plans are left exactly as they were
                raise capi.NonFiniteError(capi.ERR_NONFINITE, f"rd_loss / gradient norm is not finite (bpp {bpp}, mse {mse}, |g| {norm}); "
                                          "the optimizer step was skipped")
            gscale = inv_world
            if clip is not None and norm > clip:
                gscale *= clip / norm
            cfg = m._optimizer_config
            ops.adam_step(self.store.param, self.store.grad, self.store.m, self.store.v, lr, self.step_count + 1,
                          cfg.get("beta_1", 0.9), cfg.get("beta_2", 0.999), cfg.get("epsilon", 1e-7), grad_scale=gscale)
            self._refresh()
        self.step_count += 1
        m._step = self.step_count
        psnr = float(np.mean(-10.0 * (np.log(mse_i) - 2.0 * math.log(255.0)) / math.log(10.0)))
        return dict(rd_loss=loss, bpp=bpp, mse=mse, psnr=psnr, scheduled_lr=lr, sched_rd_lambda=rd_lambda, grad_norm=norm)

    # ---- weights out ---------------------------------------------------------------------------------
    def export_weights(self):
        """Current variables in ``Model.get_weights()`` naming (effective GDN parameters, separate base / res kernels)."""
        raw = self.store.export()
        out = OrderedDict()
        t = self.m._synthesis
        for k, v in raw.items():
            if self._two_layer and k == "synthesis/up/kernel":
                n1, nr, _ = t._names
                out[f"synthesis/{n1}/kernel"] = v[:, :, :t._ch].copy()
                if nr:
                    out[f"synthesis/{nr}/kernel"] = v[:, :, t._ch:].copy()
            elif self._two_layer and k == "synthesis/up/bias":
                n1, nr, _ = t._names
                out[f"synthesis/{n1}/bias"] = v[:t._ch].copy()
                if nr:
                    out[f"synthesis/{nr}/bias"] = v[t._ch:].copy()
            elif k == "synthesis/act/beta_raw":
                out["synthesis/act/beta"] = gdn_effective(v, GDN_BETA_MIN)
            elif k == "synthesis/act/gamma_raw":
                out["synthesis/act/gamma"] = gdn_effective(v, 0.0)
            elif k in self._reparam:
                name, kind = self._reparam[k]
                if kind == "rdft":
                    from .common.tf_checkpoint import rdft_to_kernel
                    shp = tuple(self.m.get_weights()[name].shape)
                    out[name] = rdft_to_kernel(v, shp[:2], shp[2], shp[3])
                else:
                    out[name] = gdn_effective(v, GDN_BETA_MIN if kind == "beta" else 0.0)
            else:
                out[k] = v
        order = list(self.m.get_weights())                     # Model.get_weights() order (the store is in backward order)
        return OrderedDict((k, out[k]) for k in order)

    def sync_model(self):
        """Load the trained variables into the inference ``Model`` (validation / checkpoint)."""
        self.m.set_weights(self.export_weights(), _from_trainer=True)

    def save_checkpoint(self, workdir, model_config=None, max_to_keep=1):
        """``workdir/train/checkpoints/ckpt-<step>`` in the reference's TensorBundle layout (+ ``config.json``), i.e. what
        ``common/eval_lib.py:load_latest_ckpt`` -- here and in the reference -- restores (train_lib.py:123-126,248-250,326-336).
        Variables only; the Adam moments stay in this process."""
        import json
        from pathlib import Path
        from .common import eval_lib, tf_checkpoint
        ckdir = Path(workdir) / eval_lib.TRAIN_COLLECTION / eval_lib.CHECKPOINTS_DIR_NAME
        ckdir.mkdir(parents=True, exist_ok=True)
        cfg = dict(model_config) if model_config is not None else dict(
            scheduled_num_steps=self.m._scheduled_num_steps, rd_lambda=self.m._rd_lambda, transform_config=self.m._transform_config,
            optimizer_config=self.m._optimizer_config, latent_config=self.m._latent_config)
        (Path(workdir) / "config.json").write_text(json.dumps(dict(model_config=cfg), indent=1, default=lambda o: list(o)))
        prefix = tf_checkpoint.save_reference_checkpoint(ckdir / f"ckpt-{self.step_count}", self.export_weights(),
                                                         self.m._transform_config, self.step_count)
        # optimizer state for THIS build's resume (the reference keeps iterations + Adam m / v as slot variables inside the
        # bundle; here they sit next to it): the flat moment buffers in the store's own order, keyed by its layout
        np.savez(ckdir / f"ckpt-{self.step_count}.optimizer.npz", step=np.int64(self.step_count),
                 m=self.store.m.cpu().numpy(), v=self.store.v.cpu().numpy(),
                 layout=np.array(json.dumps({k: [int(o), list(map(int, shp))] for k, (o, shp) in self.store.offsets.items()})))
        # tf.train.CheckpointManager's state file: what tf.train.latest_checkpoint (reference eval_lib.py:42-44) reads.
        # CheckpointManager(max_to_keep=train_eval_config.get('max_ckpts_to_keep', 1)), train_lib.py:124-126, keeps by RECENCY:
        # the checkpoint just written always stays (also when the directory holds bundles with higher step numbers: a reused
        # workdir, a step reset), next to the newest N - 1 others; only files this naming scheme wrote (ckpt-<digits>.*) are ever
        # removed, and only AFTER the state file names the survivors (a crash in between leaves stale files, never a state file
        # that points at deleted ones).
        import os
        import re
        name = f"ckpt-{self.step_count}"
        files = [(f, int(m.group(1))) for f in ckdir.glob("ckpt-*") if (m := re.fullmatch(r"ckpt-(\d+)\..+", f.name))]
        age = {}
        for f, k in files:
            age[k] = max(age.get(k, 0.0), f.stat().st_mtime)
        keep = checkpoints_to_keep(age, self.step_count, max_to_keep)
        state = f'model_checkpoint_path: "{name}"\n' + "".join(f'all_model_checkpoint_paths: "ckpt-{k}"\n' for k in keep)
        tmp = ckdir / "checkpoint.tmp"
        tmp.write_text(state)
        os.replace(tmp, ckdir / "checkpoint")                  # the state file changes atomically, after the bundle is complete
        for f, k in files:
            if k not in keep:
                f.unlink()
        return prefix

    def restore_optimizer(self, prefix):
        """Adam moments + iteration count written by ``save_checkpoint`` next to ``prefix``; False when there are none (a
        checkpoint of the reference, or variables only): training then continues with fresh moments at the stored step."""
        import json
        from pathlib import Path
        f = Path(str(prefix) + ".optimizer.npz")
        if not f.exists():
            return False
        d = np.load(f)
        layout = {k: [int(o), list(map(int, shp))] for k, (o, shp) in self.store.offsets.items()}
        if json.loads(str(d["layout"])) != layout:
            raise ValueError(f"{f}: optimizer state was written for a different model layout")
        self.store.m.copy_(torch.from_numpy(d["m"]).to(self.device))
        self.store.v.copy_(torch.from_numpy(d["v"]).to(self.device))
        self.step_count = int(d["step"])
        self.m._step = self.step_count
        return True
