"""TensorBundle reader + reference-Model variable mapping on hand-built bundles (CPU).  No TensorFlow-written
checkpoint exists in the build environment; these tests pin the container format as published (LevelDB table,
masked CRC32C, BundleEntryProto, string tensors, TrackableObjectGraph) and the graph walk."""
import struct

import numpy as np
import pytest

from shallow_ntc_amd.common import tf_checkpoint as ck


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert ck.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert ck.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert ck.crc32c(bytes(range(32))) == 0x46DD794E
    assert ck.crc32c(b"123456789") == 0xE3069283
    assert ck.mask_crc(0) == 0xA282EAD8


def test_varint_and_proto_roundtrip():
    for v in [0, 1, 127, 128, 300, 2**31, 2**63 + 5]:
        b = ck._put_varint(v)
        assert ck._get_varint(b, 0) == (v, len(b))
    msg = ck._field(1, 0, 150) + ck._field(2, 2, b"testing") + ck._field(6, 5, 0xDEADBEEF)
    assert ck.parse_proto(msg) == [(1, 0, 150), (2, 2, b"testing"), (6, 5, 0xDEADBEEF)]
    assert msg[:3] == b"\x08\x96\x01"          # protobuf docs: field 1 = 150


def test_snappy_decompress():
    # literal "abcd" + copy(offset 4, len 8) -> "abcdabcdabcd"; preamble = uncompressed length 12
    comp = bytes([12, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])
    assert ck.snappy_decompress(comp) == b"abcdabcdabcd"


def test_table_and_bundle_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {f"model/layer_{i}/kernel{ck.VAR_SUFFIX}": rng.standard_normal((3, 3, 4, 5)).astype(np.float32) for i in range(40)}
    tensors["step"] = np.array(1234567, np.int64)
    tensors["mask"] = np.array([True, False, True])
    tensors["half"] = rng.standard_normal((2, 3)).astype(np.float16)
    tensors[ck.OBJECT_GRAPH_KEY] = b"\x0a\x00"
    prefix = tmp_path / "ckpt-7"
    ck.write_bundle(prefix, tensors)
    idx = (tmp_path / "ckpt-7.index").read_bytes()
    assert struct.unpack("<Q", idx[-8:])[0] == ck.TABLE_MAGIC and (tmp_path / "ckpt-7.data-00000-of-00001").exists()
    back = ck.read_bundle(prefix, verify_crc=True)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        if isinstance(v, bytes):
            assert back[k] == v
        else:
            assert back[k].dtype == v.dtype and back[k].shape == v.shape
            np.testing.assert_array_equal(back[k], v)
    # corruption is detected: flip one data byte, then one index byte
    data = bytearray((tmp_path / "ckpt-7.data-00000-of-00001").read_bytes())
    data[10] ^= 0xFF
    (tmp_path / "ckpt-7.data-00000-of-00001").write_bytes(bytes(data))
    with pytest.raises(ValueError, match="checksum"):
        ck.read_bundle(prefix, verify_crc=True)
    idx = bytearray(idx)
    idx[5] ^= 0x01
    (tmp_path / "ckpt-7.index").write_bytes(bytes(idx))
    with pytest.raises(ValueError):
        ck.read_bundle(prefix)
    (tmp_path / "bad.index").write_bytes(b"\x00" * 64)
    with pytest.raises(ValueError, match="magic"):
        ck.read_table(tmp_path / "bad.index")


def test_gdn_reparameterisation():
    beta = np.array([1.0, 1e-6, 0.5, 3.0], np.float32)
    np.testing.assert_allclose(ck.gdn_parameter_value(ck.gdn_parameter_variable(beta), 1e-6), beta, rtol=1e-6)
    gamma = np.array([[0.1, 0.0], [0.02, 0.1]], np.float32)
    np.testing.assert_allclose(ck.gdn_parameter_value(ck.gdn_parameter_variable(gamma), 0.0), gamma, atol=1e-9)
    assert ck.gdn_parameter_value(np.array([0.0]), 1e-6)[0] == pytest.approx(1e-6, rel=1e-3)      # lower bound kicks in


class _GraphBuilder:
    """Builds a TrackableObjectGraph + tensors shaped like tf.train.Checkpoint(model=Model) of the reference."""

    def __init__(self):
        self.nodes = [({}, {})]
        self.tensors = {}

    def add(self, parent, name):
        self.nodes.append(({}, {}))
        self.nodes[parent][0][name] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def var(self, parent, name, path, value):
        n = self.add(parent, name)
        key = path + "/" + name + ck.VAR_SUFFIX
        self.nodes[n][1]["VARIABLE_VALUE"] = key
        self.tensors[key] = value
        return n

    def conv(self, parent, name, path, w, prefix):
        n = self.add(parent, name)
        self.var(n, "kernel", f"{path}/{name}", w[prefix + "/kernel"])
        if prefix + "/bias" in w:
            self.var(n, "bias", f"{path}/{name}", w[prefix + "/bias"])
        return n

    def rb(self, parent, name, path, w, prefix):
        n = self.add(parent, name)
        blk = self.add(n, "_block")
        for i in range(3):
            self.conv(blk, f"layer_with_weights-{i}", f"{path}/{name}/_block", w, f"{prefix}/conv{i}")
        return n


def _reference_like_checkpoint(tmp_path, weights):
    b = _GraphBuilder()
    model = b.add(0, "model")
    b.add(model, "optimizer")
    ana = b.add(model, "_analysis")
    seq = b.add(ana, "_transform")
    order = ["conv0", "rb0", "rb1", "rb2", "conv1", "rb3", "rb4", "rb5", "attn0", "conv2", "rb6", "rb7", "rb8", "conv3", "attn1"]
    for i, item in enumerate(order):
        lw, path, pre = f"layer_with_weights-{i}", "model/_analysis/_transform", f"analysis/{item}"
        if item.startswith("conv"):
            b.conv(seq, lw, path, weights, pre)
        elif item.startswith("rb"):
            b.rb(seq, lw, path, weights, pre)
        else:
            att = b.add(seq, lw)
            trunk, branch = b.add(att, "_trunk"), b.add(att, "_attention_branch")
            b.add(att, "_branch_layers")
            for j in range(3):
                b.rb(trunk, f"layer_with_weights-{j}", f"{path}/{lw}/_trunk", weights, f"{pre}/trunk/rb{j}")
                b.rb(branch, f"layer_with_weights-{j}", f"{path}/{lw}/_attention_branch", weights, f"{pre}/branch/rb{j}")
            b.conv(branch, "layer_with_weights-3", f"{path}/{lw}/_attention_branch", weights, f"{pre}/branch/conv")
    syn = b.add(model, "_synthesis")
    for n in ("base_conv", "res", "out_conv"):
        b.conv(syn, n, "model/_synthesis", weights, f"synthesis/{n}")
    act = b.add(syn, "activation")
    for pname, ours in (("beta_parameter", "beta"), ("gamma_parameter", "gamma")):
        p = b.add(act, pname)
        b.var(p, "variable", f"model/_synthesis/activation/{pname}", ck.gdn_parameter_variable(weights[f"synthesis/act/{ours}"]))
    for tname in ("_hyper_analysis", "_hyper_synthesis"):
        t = b.add(model, tname)
        for i in range(3):
            b.conv(t, f"layer_with_weights-{i}", f"model/{tname}", weights, f"{tname[1:]}/layer_{i}")
    prior = b.add(model, "_prior")
    base = b.add(prior, "_base")
    for kind, ours in (("_matrices", "matrix"), ("_biases", "bias"), ("_factors", "factor")):
        lst = b.add(base, kind)
        i = 0
        while f"prior/{ours}_{i}" in weights:
            v = weights[f"prior/{ours}_{i}"]
            b.var(lst, str(i), f"model/_prior/_base/{kind}", v if ours == "matrix" else v[..., None])
            i += 1
    tensors = dict(b.tensors)
    tensors[ck.OBJECT_GRAPH_KEY] = ck.ObjectGraph.serialize(b.nodes)
    tensors["save_counter" + ck.VAR_SUFFIX] = np.array(3, np.int64)
    ck.write_bundle(tmp_path / "ckpt-3", tensors)
    return tmp_path / "ckpt-3"


def test_reference_model_mapping(tmp_path):
    """A bundle laid out like the reference's checkpoint maps onto exactly the variable inventory the oracle /
    product Model use (names + shapes), GDN parameters are un-reparameterised, prior biases lose their
    trailing unit axis."""
    from oracle import model_np
    tc = dict(analysis=dict(cls="ElicAnalysis", channels=(8, 8, 8, 16)),
              synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                             activation_type="igdn", res_type="conv"))
    ref = model_np.Model(tc)
    rng = np.random.default_rng(1)
    w = {k: (rng.standard_normal(v.shape) * 0.1).astype(np.float32) for k, v in ref.init_params(seed=5).items()}
    w["synthesis/act/beta"] = (1 + rng.random(12)).astype(np.float32)
    w["synthesis/act/gamma"] = (0.1 * np.eye(12) + 0.01 * rng.random((12, 12))).astype(np.float32)
    prefix = _reference_like_checkpoint(tmp_path, w)
    got = ck.load_reference_checkpoint(prefix, tc)
    assert set(got) == set(ref.param_shapes())
    for k, shp in ref.param_shapes().items():
        assert tuple(got[k].shape) == tuple(shp), k
        np.testing.assert_allclose(got[k], w[k], rtol=2e-6, atol=1e-9)
    # the walk fails loudly, naming what it saw, when the graph differs
    with pytest.raises(KeyError, match="children"):
        ck.load_reference_checkpoint(prefix, dict(tc, synthesis=dict(cls="JPEGLikeSynthesis")))
    with pytest.raises(KeyError):               # an ELIC graph read as a SignalConv2D stack: wrong layer count / no rdft variable
        ck.load_reference_checkpoint(prefix, dict(tc, analysis=dict(cls="MBT2018Analysis", channels_base=8)))
    with pytest.raises(NotImplementedError):
        ck.load_reference_checkpoint(prefix, dict(tc, analysis=dict(cls="NoSuchAnalysis")))


@pytest.mark.parametrize("analysis,synthesis", [
    (dict(cls="ElicAnalysis", channels=(8, 8, 8, 16)), dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn")),
    (dict(cls="CNNAnalysis", channels_base=8, output_channels=16), dict(cls="TwoLayerSynthesis", channels=(24, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn")),
    (dict(cls="ElicAnalysis", channels=(8, 8, 8, 16)), dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16)),
])
def test_writer_is_the_inverse_of_the_reader(tmp_path, analysis, synthesis):
    """save_reference_checkpoint -> load_reference_checkpoint returns every variable (the GDN parameters through tfc's
    reparameterisation and back); the writer produces the same keys as the hand-built reference-shaped bundle above."""
    from oracle import model_np
    cfg = dict(analysis=analysis, synthesis=synthesis)
    m = model_np.Model(cfg)
    w = m.init_params(5)
    prefix = ck.save_reference_checkpoint(tmp_path / "ckpt-7", w, cfg, step=7)
    back = ck.load_reference_checkpoint(prefix, cfg)
    assert set(back) == set(w)
    for k in w:
        np.testing.assert_allclose(back[k], w[k], rtol=0, atol=2e-7 if "/act/" in k else 0, err_msg=k)
    assert int(ck.read_bundle(prefix)["save_counter" + ck.VAR_SUFFIX]) == 7
    if synthesis["cls"] == "TwoLayerResSynthesis":
        (tmp_path / "hand").mkdir()
        hand = ck.read_bundle(_reference_like_checkpoint(tmp_path / "hand", w))
        ours = ck.read_bundle(prefix)
        assert {k for k in hand if k.endswith(ck.VAR_SUFFIX)} == {k for k in ours if k.endswith(ck.VAR_SUFFIX)}


def test_rdft_basis_is_a_parseval_frame_and_round_trips_kernels():
    """tfc.RDFTParameter (default kernel_parameter of tfc.SignalConv2D; reference common/transforms.py:101-175): the basis
    M satisfies M M^T = I for every kernel shape the reference uses, kernel -> rdft -> kernel is the identity, and the
    DC coefficient is the kernel sum / sqrt(size).  [Restated from the published definition; unverified against TF.]"""
    from shallow_ntc_amd.common import tf_checkpoint as tc
    rng = np.random.default_rng(0)
    for shape in ((5, 5), (9, 9), (3, 3), (4, 6), (1, 1), (2, 5)):
        m = tc.irdft_matrix(shape)
        size = shape[0] * shape[1]
        assert m.shape == (size, 2 * shape[0] * (shape[1] // 2 + 1))
        np.testing.assert_allclose(m @ m.T, np.eye(size), atol=1e-12)
        k = rng.standard_normal(shape + (3, 4)).astype(np.float32)
        r = tc.kernel_to_rdft(k)
        assert r.shape == (m.shape[1], 12)
        np.testing.assert_allclose(tc.rdft_to_kernel(r, shape, 3, 4), k, atol=2e-6)
        np.testing.assert_allclose(r[0], k.reshape(size, -1).sum(0) / np.sqrt(size), rtol=1e-5, atol=1e-6)   # DC term
    # a pure cosine along x lives in exactly one real coefficient: the basis really is the DFT, not just any frame
    yy, xx = np.mgrid[0:5, 0:5]
    k = np.cos(2 * np.pi * xx / 5.0)[..., None, None].astype(np.float32)
    r = tc.kernel_to_rdft(k)[:, 0]
    assert (np.abs(r) > 1e-5).sum() == 1 and abs(np.abs(r).max() - np.sqrt(25 / 2)) < 1e-4
    with pytest.raises(ValueError):
        tc.rdft_to_kernel(np.zeros((25, 12), np.float32), (5, 5), 3, 4)          # square layout: not TFC's frame
    # the two published-definition readings of the column order are permutations of each other (same frame, same shape):
    # nothing but a TensorFlow-written bundle can tell them apart, which is why import / export warn (next test)
    for shape in ((5, 5), (9, 9)):
        a, b = tc.irdft_matrix(shape, "real_then_imag"), tc.irdft_matrix(shape, "interleaved")
        assert a.shape == b.shape and not np.allclose(a, b)
        np.testing.assert_allclose(a @ a.T, b @ b.T, atol=1e-12)
        half = shape[1] // 2 + 1
        perm = [(i // half) * 2 * half + (i % half) for i in range(shape[0] * half)]
        perm += [p + half for p in perm]
        np.testing.assert_allclose(a, b[:, perm], atol=0)
    assert tc.RDFT_LAYOUT in tc.RDFT_LAYOUTS
    with pytest.raises(ValueError):
        tc.irdft_matrix((5, 5), "columns")


@pytest.mark.parametrize("which", ["mbt2018", "bls2017"])
def test_signal_conv_checkpoints_round_trip(which, tmp_path):
    """BASELINE configs 1 and 2: a hand-built TensorBundle whose SignalConv2D kernels are stored as RDFT variables and
    whose GDN parameters are reparameterised comes back as the effective weights (object-graph walk of the Sequential
    stacks of reference common/transforms.py:93-175; bls2017 is the factorized-prior model without hyper transforms)."""
    from oracle import model_np
    from shallow_ntc_amd.common import tf_checkpoint as tc
    if which == "mbt2018":
        tconf = dict(analysis=dict(cls="MBT2018Analysis", channels_base=8, output_channels=12),
                     synthesis=dict(cls="MBT2018Synthesis", channels_base=8))
        ref = model_np.Model(tconf)
    else:
        tconf = dict(analysis=dict(cls="BLS2017Analysis", num_filters=8), synthesis=dict(cls="BLS2017Synthesis", num_filters=8))
        ref = model_np.Model(tconf, factorized=True)
    w = ref.init_params(5)
    rng = np.random.default_rng(1)
    for k in w:                                  # non-trivial GDN parameters and biases
        if k.endswith("/beta"):
            w[k] = (1 + rng.random(w[k].shape)).astype(np.float32)
        elif k.endswith("/gamma"):
            w[k] = (0.1 * rng.random(w[k].shape)).astype(np.float32)
        elif k.endswith("/bias"):
            w[k] = rng.standard_normal(w[k].shape).astype(np.float32)
    tc._rdft_warned.clear()
    with pytest.warns(RuntimeWarning, match="NOT been verified against a TensorFlow-written bundle"):     # loud, once per direction
        prefix = tc.save_reference_checkpoint(tmp_path / "ckpt-7", w, tconf, step=7)
    raw = tc.read_bundle(prefix)
    rdft_keys = [k for k in raw if "/rdft/" in k]
    assert len(rdft_keys) == sum(1 for k in w if k.endswith("/kernel") and k.split("/")[0] in ("analysis", "synthesis"))
    assert all(raw[k].shape[0] in (30, 90) for k in rdft_keys)          # 5x5 -> 2*5*3 rows, 9x9 -> 2*9*5 rows
    # this repository's writer records the rdft column order it used; the reader honours the record (no warning: the order is known)
    assert tc.RDFT_LAYOUTS[int(raw[tc.WRITER_LAYOUT_KEY])] == tc.RDFT_LAYOUT
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = tc.load_reference_checkpoint(prefix, tconf)
    assert set(got) == set(w)
    for k in w:
        np.testing.assert_allclose(got[k], w[k], rtol=2e-6, atol=2e-6, err_msg=k)
    # ... also when the module default has changed since the bundle was written
    other = [l for l in tc.RDFT_LAYOUTS if l != tc.RDFT_LAYOUT][0]
    saved = tc.RDFT_LAYOUT
    try:
        tc.RDFT_LAYOUT = other
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            again = tc.load_reference_checkpoint(prefix, tconf)
    finally:
        tc.RDFT_LAYOUT = saved
    for k in w:
        np.testing.assert_allclose(again[k], w[k], rtol=2e-6, atol=2e-6, err_msg=k)
    # a bundle without the record is TensorFlow's as far as anyone can tell: module default, loud warning ...
    bare = {k: v for k, v in raw.items() if k != tc.WRITER_LAYOUT_KEY}
    tc.write_bundle(tmp_path / "bare-7", bare)
    tc._rdft_warned.clear()
    with pytest.warns(RuntimeWarning, match="import of tfc.SignalConv2D kernels"):
        tc.load_reference_checkpoint(tmp_path / "bare-7", tconf)
    # ... unless the side file of this repository's trainer says an EARLIER build wrote it: refused, the order is unknown
    (tmp_path / "bare-7.optimizer.npz").write_bytes(b"")
    with pytest.raises(ValueError, match="earlier build"):
        tc.load_reference_checkpoint(tmp_path / "bare-7", tconf)
