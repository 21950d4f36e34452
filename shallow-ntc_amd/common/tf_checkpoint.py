"""TensorFlow checkpoint (TensorBundle) reader + the variable mapping of the reference's Model
(reference common/eval_lib.py:11-53, common/train_lib.py:123-126).

Pure Python + NumPy, no TensorFlow.  ``tf.train.Checkpoint(model=model).save`` writes

    ckpt-N.index                 a LevelDB-format table: "" -> BundleHeaderProto, key -> BundleEntryProto
    ckpt-N.data-00000-of-00001   raw little-endian tensor bytes at (offset, size)

and one string tensor ``_CHECKPOINTABLE_OBJECT_GRAPH`` (a TrackableObjectGraph proto) that records the
Python object graph; variables are addressed through it (``children`` by attribute name, Keras layers of
a Sequential as ``layer_with_weights-i``), so the mapping below walks the *graph*, not key strings.

STATUS: the container format (table blocks, varints, masked CRC32C, protos, string tensors) is implemented
from its published layout and exercised on bundles written by ``write_bundle`` in this file; no checkpoint
produced by TensorFlow itself was available in the build environment (no TF, no network, the trained
checkpoints are an external download -- reference README.md:21,106).  ``load_reference_checkpoint`` fails
loudly, listing the children it saw, wherever the object graph differs from what the reference source implies.
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from pathlib import Path

import os
import warnings

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
VAR_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          19: np.float16, 17: np.uint16, 22: np.uint32, 23: np.uint64}
DT_STRING = 7
_DT_OF = {np.dtype(v): k for k, v in DTYPES.items()}


# ------------------------------------------------------------------------------------------ crc32c
def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC = _crc_table()


def crc32c(data: bytes, crc=0) -> int:
    """CRC-32C (Castagnoli), byte-wise table walk (pure Python: ~1 MB/s, see read_bundle(verify_crc=...))."""
    crc ^= 0xFFFFFFFF
    tbl = _CRC
    for b in data:
        crc = tbl[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask_crc(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------ varints / protos
def _get_varint(buf, pos):
    r, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def parse_proto(buf) -> list:
    """Minimal protobuf wire parser -> [(field, wire_type, value)]; length-delimited values stay bytes."""
    out, pos = [], 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((f, wt, v))
    return out


def _field(f, wt, payload) -> bytes:
    tag = _put_varint((f << 3) | wt)
    if wt == 0:
        return tag + _put_varint(payload)
    if wt == 2:
        return tag + _put_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wt)


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


# ------------------------------------------------------------------------------------------ snappy (decode only)
def snappy_decompress(buf: bytes) -> bytes:
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------ LevelDB table
def _read_block(data: bytes, offset: int, size: int, verify=True) -> bytes:
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if verify and mask_crc(crc32c(raw + bytes([ctype]))) != stored:
        raise ValueError("table block checksum mismatch")
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise ValueError(f"unknown block compression {ctype}")


def _block_entries(block: bytes):
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path) -> "OrderedDict[bytes, bytes]":
    data = Path(path).read_bytes()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorBundle index (bad table magic)")
    footer = data[-48:]
    _, p = _get_varint(footer, 0)            # metaindex handle (unused)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    out = OrderedDict()
    for _, handle in _block_entries(_read_block(data, ioff, isize)):
        boff, q = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, q)
        for k, v in _block_entries(_read_block(data, boff, bsize)):
            out[k] = v
    return out


def _build_block(entries) -> bytes:
    body = bytearray()
    for k, v in entries:                        # restart interval 1: no prefix sharing, every entry is a restart
        body += _put_varint(0) + _put_varint(len(k)) + _put_varint(len(v)) + k + v
    restarts, pos = [], 0
    for k, v in entries:
        restarts.append(pos)
        pos += len(_put_varint(0)) + len(_put_varint(len(k))) + len(_put_varint(len(v))) + len(k) + len(v)
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def write_table(path, items: "list[tuple[bytes, bytes]]", entries_per_block=16):
    items = sorted(items)
    out = bytearray()
    index = []

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)

    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        off, size = emit(_build_block(chunk))
        index.append((chunk[-1][0], _put_varint(off) + _put_varint(size)))
    moff, msize = emit(_build_block([]))
    ioff, isize = emit(_build_block(index))
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    Path(path).write_bytes(bytes(out))


# ------------------------------------------------------------------------------------------ bundle
def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, crc=None)
    for f, wt, v in parse_proto(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _, v2 in parse_proto(v):
                if f2 == 2:
                    dim = dict((a, c) for a, _, c in parse_proto(v2))
                    e["shape"].append(_signed64(dim.get(1, 0)))
        elif f == 3:
            e["shard"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc"] = v
        elif f == 7:
            raise NotImplementedError("sliced (partitioned) variables are not used by the reference")
    return e


def read_bundle(prefix, verify_crc="auto") -> "OrderedDict[str, np.ndarray | bytes]":
    """{key: ndarray} for every tensor of the checkpoint ``prefix`` (e.g. '.../checkpoints/ckpt-180');
    string tensors come back as bytes (scalar) or lists of bytes.  verify_crc: True = every tensor, "auto" =
    tensors up to 64 KiB (the pure-Python CRC is slow; table blocks are always verified), False = none."""
    prefix = str(prefix)
    table = read_table(prefix + ".index")
    header = dict((f, v) for f, _, v in parse_proto(table.get(b"", b"")))
    num_shards = header.get(1, 1)
    if header.get(2, 0) != 0:
        raise NotImplementedError("big-endian bundles")
    shards = {}
    out = OrderedDict()
    for key, val in table.items():
        if key == b"":
            continue
        e = _parse_entry(val)
        if e["shard"] not in shards:
            shards[e["shard"]] = Path(f"{prefix}.data-{e['shard']:05d}-of-{num_shards:05d}").read_bytes()
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError(f"{key!r}: data shard is truncated")
        name = key.decode()
        if e["dtype"] == DT_STRING:
            n = int(np.prod(e["shape"])) if e["shape"] else 1
            pos, lens = 0, []
            for _ in range(n):
                ln, pos = _get_varint(raw, pos)
                lens.append(ln)
            pos += 4                                        # masked crc32c of the length varints
            strs = []
            for ln in lens:
                strs.append(bytes(raw[pos:pos + ln]))
                pos += ln
            out[name] = strs[0] if not e["shape"] else strs
            continue
        if e["dtype"] not in DTYPES:
            raise NotImplementedError(f"{name}: DataType {e['dtype']}")
        check = verify_crc is True or (verify_crc == "auto" and len(raw) <= 65536)
        if check and e["crc"] is not None and mask_crc(crc32c(raw)) != e["crc"]:
            raise ValueError(f"{name}: tensor checksum mismatch")
        out[name] = np.frombuffer(raw, dtype=np.dtype(DTYPES[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    return out


def write_bundle(prefix, tensors: dict):
    """Writer used by the tests to hand-build bundles (single shard, no compression)."""
    prefix = str(prefix)
    data = bytearray()
    items = [(b"", _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1)))]      # num_shards = 1, version.producer = 1
    for name in sorted(tensors):
        v = tensors[name]
        off = len(data)
        if isinstance(v, (bytes, str)):
            b = v.encode() if isinstance(v, str) else v
            lens = _put_varint(len(b))
            data += lens + struct.pack("<I", mask_crc(crc32c(lens))) + b
            entry = _field(1, 0, DT_STRING) + _field(2, 2, b"") + _field(4, 0, off) + _field(5, 0, len(data) - off)
            entry += _field(6, 5, mask_crc(crc32c(bytes(data[off:]))))
        else:
            a = np.asarray(v)                    # (ascontiguousarray would turn a scalar into shape (1,))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            data += raw
            shape = b"".join(_field(2, 2, _field(1, 0, int(d))) for d in a.shape)
            entry = _field(1, 0, _DT_OF[a.dtype]) + _field(2, 2, shape) + _field(4, 0, off) + _field(5, 0, len(raw))
            entry += _field(6, 5, mask_crc(crc32c(raw)))
        items.append((name.encode(), entry))
    write_table(prefix + ".index", items)
    Path(prefix + ".data-00000-of-00001").write_bytes(bytes(data))


# ------------------------------------------------------------------------------------------ object graph
class ObjectGraph:
    """TrackableObjectGraph: nodes[i] = (children {local_name: node_id}, attributes {name: checkpoint_key})."""

    def __init__(self, blob: bytes):
        self.children, self.attrs = [], []
        for f, _, node in parse_proto(blob):
            if f != 1:
                continue
            ch, at = OrderedDict(), OrderedDict()
            for f2, _, v in parse_proto(node):
                if f2 == 1:
                    d = dict((a, c) for a, _, c in parse_proto(v))
                    ch[d.get(2, b"").decode()] = d.get(1, 0)
                elif f2 == 2:
                    d = dict((a, c) for a, _, c in parse_proto(v))
                    at[d.get(1, b"").decode()] = d.get(3, b"").decode()
            self.children.append(ch)
            self.attrs.append(at)

    def child(self, node, *names):
        """Follow the first existing local name among ``names`` (alternatives for private/public spellings)."""
        for n in names:
            if n in self.children[node]:
                return self.children[node][n]
        raise KeyError(f"none of {names} among children {list(self.children[node])} of node {node}")

    def has(self, node, name):
        return name in self.children[node]

    def path(self, node, *path):
        for p in path:
            node = self.child(node, *(p if isinstance(p, tuple) else (p,)))
        return node

    def variable_key(self, node):
        at = self.attrs[node]
        if "VARIABLE_VALUE" not in at:
            raise KeyError(f"node {node} is not a variable (attributes {list(at)}, children {list(self.children[node])})")
        return at["VARIABLE_VALUE"]

    def layers_with_weights(self, node):
        out, i = [], 0
        while f"layer_with_weights-{i}" in self.children[node]:
            out.append(self.children[node][f"layer_with_weights-{i}"])
            i += 1
        return out

    @staticmethod
    def serialize(nodes) -> bytes:
        """nodes: list of (children {name: id}, attributes {name: checkpoint_key}) -> proto bytes (tests)."""
        out = b""
        for ch, at in nodes:
            body = b"".join(_field(1, 2, _field(1, 0, i) + _field(2, 2, n.encode())) for n, i in ch.items())
            body += b"".join(_field(2, 2, _field(1, 2, n.encode()) + _field(3, 2, k.encode())) for n, k in at.items())
            out += _field(1, 2, body)
        return out


# ------------------------------------------------------------------------------------------ TFC re-parameterisations
GDN_PEDESTAL = 2.0 ** -36                    # tfc GDNParameter: offset = 2^-18, pedestal = offset^2


def gdn_parameter_value(variable, minimum):
    """tfc.layers.GDNParameter: value = max(variable, sqrt(minimum + pedestal))^2 - pedestal
    (beta: minimum 1e-6, gamma: minimum 0)."""
    v = np.asarray(variable, np.float64)
    bound = np.sqrt(minimum + GDN_PEDESTAL)
    return (np.maximum(v, bound) ** 2 - GDN_PEDESTAL).astype(np.float32)


def gdn_parameter_variable(value):
    """Inverse (what TFC stores for a given effective value): sqrt(max(value + pedestal, pedestal))."""
    return np.sqrt(np.maximum(np.asarray(value, np.float64) + GDN_PEDESTAL, GDN_PEDESTAL)).astype(np.float32)


RDFT_LAYOUTS = ("real_then_imag", "interleaved")
RDFT_LAYOUT = os.environ.get("SNTC_RDFT_LAYOUT", "real_then_imag")
_rdft_warned = set()


def _warn_rdft(direction):
    """Import / export of tfc.RDFTParameter variables is UNVERIFIED against a TensorFlow-written checkpoint: say so, loudly,
    once per direction and process.  Both column orders of the basis are orthonormal frames of the same shape, so no shape or
    round-trip check can tell them apart -- only a real SignalConv2D checkpoint can."""
    if direction not in _rdft_warned:
        _rdft_warned.add(direction)
        warnings.warn(f"tf_checkpoint: {direction} of tfc.SignalConv2D kernels stored as real-DFT coefficients (bls2017 / mbt2018 "
                      f"models) assumes the '{RDFT_LAYOUT}' column order of tfc's irdft_matrix; this restatement has NOT been "
                      "verified against a TensorFlow-written bundle -- a wrong order imports wrong kernels silently.  Check one "
                      "layer against the reference (or set SNTC_RDFT_LAYOUT to the other order, 'interleaved' / 'real_then_imag') "
                      "before trusting the result.", RuntimeWarning, stacklevel=3)


def irdft_matrix(shape, layout=None):
    """tfc's real-DFT kernel basis (tensorflow_compression spectral_ops.irdft_matrix, the matrix behind
    ``tfc.layers.RDFTParameter``, the default ``kernel_parameter="rdft"`` of tfc.SignalConv2D; reference call sites
    common/transforms.py:101-112,123-134,152-155,172-175) [DEP: restated from the published definition, NOT verified
    against a TF-written checkpoint]: the rows of the identity over the kernel's spatial ``shape`` are taken through
    ``rfftn``; the bins of the last axis that have a distinct conjugate partner are scaled by sqrt(2), everything by
    1 / sqrt(size).  Column order (``layout``, default the module's RDFT_LAYOUT / SNTC_RDFT_LAYOUT):
    'real_then_imag' -- the half spectrum is flattened to [size, -1] first and then [real | imag] are concatenated (all real
    columns, then all imaginary ones; the default: tfc reshapes before it concatenates); 'interleaved' -- real and imaginary
    parts are concatenated along the last frequency axis before flattening (per frequency row [re.. | im..]; the order this
    repository used through round 2).  Either way M is [size, 2 * prod(shape[:-1]) * (shape[-1] // 2 + 1)] with
    M @ M.T = I, so that
        rdft   = M.T @ kernel.reshape(size, cin * cout)         (what the checkpoint stores)
        kernel = (M @ rdft).reshape(*shape, cin, cout)          (what the layer convolves with);
    the two orders differ by a permutation of the rdft rows only (training is unaffected: Adam is element-wise)."""
    layout = layout or RDFT_LAYOUT
    if layout not in RDFT_LAYOUTS:
        raise ValueError(f"rdft layout must be one of {RDFT_LAYOUTS}, not {layout!r}")
    shape = tuple(int(v) for v in shape)
    size = int(np.prod(shape))
    rank = len(shape)
    m = np.identity(size, dtype=np.float64).reshape((size,) + shape)
    f = np.fft.rfftn(m, axes=tuple(range(1, rank + 1)))
    n = shape[-1]
    f[..., 1:(n + 1) // 2] *= np.sqrt(2.0)
    f /= np.sqrt(size)
    if layout == "interleaved":
        return np.concatenate([f.real, f.imag], axis=-1).reshape(size, -1)
    f = f.reshape(size, -1)
    return np.concatenate([f.real, f.imag], axis=-1)


WRITER_LAYOUT_KEY = "_sntc_writer/rdft_layout"     # int32 index into RDFT_LAYOUTS: bundles THIS repository writes say which column
                                                   # order their rdft variables use (a TensorFlow reader never asks for the key)


def rdft_to_kernel(rdft, spatial, cin, cout, layout=None):
    """RDFTParameter variable -> SignalConv2D kernel [kh, kw, cin, cout] (float32)."""
    m = irdft_matrix(spatial, layout)
    r = np.asarray(rdft, np.float64)
    r = r.reshape(r.shape[0], -1)
    if r.shape[0] != m.shape[1] or r.shape[1] != cin * cout:
        raise ValueError(f"rdft variable of shape {np.asarray(rdft).shape} does not fit a {tuple(spatial)} x {cin} x {cout} kernel "
                         f"(expected {m.shape[1]} x {cin * cout}: the half-spectrum real / imaginary layout of irdft_matrix)")
    return (m @ r).reshape(tuple(spatial) + (cin, cout)).astype(np.float32)


def kernel_to_rdft(kernel):
    """SignalConv2D kernel [kh, kw, cin, cout] -> the RDFTParameter variable [rows, cin * cout] (float32)."""
    k = np.asarray(kernel, np.float64)
    m = irdft_matrix(k.shape[:2])
    return (m.T @ k.reshape(k.shape[0] * k.shape[1], -1)).astype(np.float32)


# ------------------------------------------------------------------------------------------ reference Model mapping
class CheckpointMapper:
    """Collects {our variable name: ndarray} by walking the object graph the way the reference's classes are
    written (attribute names from reference common/elic.py, common/transforms.py, mshyper/models.py)."""

    def __init__(self, tensors, graph: ObjectGraph, prefix=None):
        self.t, self.g = tensors, graph
        self.out = OrderedDict()
        # the rdft column order: recorded by this repository's own writer; a bundle without the record is either TensorFlow's
        # (module default + loud warning) or one an EARLIER build of this repository wrote, which is refused: its order was
        # 'interleaved' through round 2 and 'real_then_imag' in round 3, and nothing in it says which
        self.rdft_layout = None
        if WRITER_LAYOUT_KEY in tensors:
            self.rdft_layout = RDFT_LAYOUTS[int(np.asarray(tensors[WRITER_LAYOUT_KEY]).reshape(-1)[0])]
        self._ours_unmarked = prefix is not None and Path(str(prefix) + ".optimizer.npz").exists() and self.rdft_layout is None

    def var(self, node):
        return self.t[self.g.variable_key(node)]

    def conv(self, node, name, bias=True):
        self.out[f"{name}/kernel"] = self.var(self.g.child(node, "kernel"))
        if bias:
            self.out[f"{name}/bias"] = self.var(self.g.child(node, "bias"))

    def signal_conv(self, node, name, spatial, cin, cout, bias=True):     # tfc.SignalConv2D, kernel_parameter="rdft"
        kp = self.g.path(node, ("_kernel_parameter", "kernel_parameter", "kernel"), "rdft")
        if self._ours_unmarked and not os.environ.get("SNTC_RDFT_LAYOUT"):
            raise ValueError("this bundle was written by an earlier build of this repository (it has an .optimizer.npz side file but no "
                             f"{WRITER_LAYOUT_KEY} record): its rdft column order is unknown ('interleaved' through round 2, "
                             "'real_then_imag' in round 3).  Set SNTC_RDFT_LAYOUT to the order it was written with to import it.")
        if self.rdft_layout is None:
            _warn_rdft("import")
        self.out[f"{name}/kernel"] = rdft_to_kernel(self.var(kp), spatial, cin, cout, self.rdft_layout)
        if bias:
            self.out[f"{name}/bias"] = self.var(self.g.child(node, "_bias_parameter", "bias_parameter", "bias"))

    def signal_stack(self, node, prefix, shapes, gdn_names):
        """tf.keras.Sequential of tfc.SignalConv2D layers whose ``activation`` is a tfc.GDN layer (MBT2018*, BLS2017*:
        transforms.py:93-175).  ``shapes``: our parameter shapes, which give every kernel's (kh, kw, cin, cout)."""
        layers = self.g.layers_with_weights(node)
        nconv = sum(1 for k in shapes if k.endswith("/kernel"))
        if len(layers) != nconv:
            raise KeyError(f"{prefix}: expected {nconv} layers with weights, found {len(layers)}")
        for i, layer in enumerate(layers):
            kh, kw, cin, cout = shapes[f"layer_{i}/kernel"]
            self.signal_conv(layer, f"{prefix}layer_{i}", (kh, kw), cin, cout, bias=f"layer_{i}/bias" in shapes)
            gname = gdn_names.format(i)
            if f"{gname}/beta" in shapes:
                act = self.g.child(layer, "_activation") if self.g.has(layer, "_activation") else self.g.child(layer, "activation")
                self.gdn1(act, prefix + gname)

    def residual_block(self, node, name):                     # elic.py:57-64: self._block = Sequential([...])
        convs = self.g.layers_with_weights(self.g.child(node, "_block"))
        if len(convs) != 3:
            raise KeyError(f"{name}: expected 3 convolutions in a ResidualBlock, found {len(convs)}")
        for i, c in enumerate(convs):
            self.conv(c, f"{name}/conv{i}")

    def attention(self, node, name):                          # elic.py:83-95
        for i, rb in enumerate(self.g.layers_with_weights(self.g.child(node, "_trunk"))):
            self.residual_block(rb, f"{name}/trunk/rb{i}")
        branch = self.g.layers_with_weights(self.g.child(node, "_attention_branch"))
        for i, rb in enumerate(branch[:-1]):
            self.residual_block(rb, f"{name}/branch/rb{i}")
        self.conv(branch[-1], f"{name}/branch/conv")

    def elic_analysis(self, node, prefix):                    # elic.py:141-163
        seq = self.g.layers_with_weights(self.g.child(node, "_transform"))
        nconv = nrb = natt = 0
        for layer in seq:
            if self.g.has(layer, "_block"):
                self.residual_block(layer, f"{prefix}rb{nrb}")
                nrb += 1
            elif self.g.has(layer, "_trunk"):
                self.attention(layer, f"{prefix}attn{natt}")
                natt += 1
            else:
                self.conv(layer, f"{prefix}conv{nconv}")
                nconv += 1

    def sequential_convs(self, node, prefix, names):          # HyperAnalysis / HyperSynthesis / CNN* (transforms.py:179-232)
        layers = self.g.layers_with_weights(node)
        if len(layers) != len(names):
            raise KeyError(f"{prefix}: expected {len(names)} layers with weights, found {len(layers)}")
        for layer, n in zip(layers, names):
            self.conv(layer, prefix + n)

    def gdn1(self, node, name):                                # tfc.GDN with trainable beta / gamma GDNParameters
        beta = self.g.path(node, ("beta_parameter", "_beta_parameter", "beta"), "variable")
        gamma = self.g.path(node, ("gamma_parameter", "_gamma_parameter", "gamma"), "variable")
        self.out[f"{name}/beta"] = gdn_parameter_value(self.var(beta), 1e-6)
        self.out[f"{name}/gamma"] = gdn_parameter_value(self.var(gamma), 0.0)

    def two_layer_res(self, node, prefix):                    # transforms.py:331-357
        self.conv(self.g.child(node, "base_conv"), prefix + "base_conv")
        self.conv(self.g.child(node, "res"), prefix + "res")
        self.conv(self.g.child(node, "out_conv"), prefix + "out_conv")
        if self.g.has(node, "activation"):
            self.gdn1(self.g.child(node, "activation"), prefix + "act")

    def two_layer(self, node, prefix):                        # transforms.py:307-313
        self.conv(self.g.child(node, "conv1"), prefix + "conv1")
        self.conv(self.g.child(node, "conv2"), prefix + "conv2")
        c1 = self.g.child(node, "conv1")
        if self.g.has(c1, "activation"):
            self.gdn1(self.g.child(c1, "activation"), prefix + "act")

    def jpeg_like(self, node, prefix):                        # transforms.py:284-287
        self.conv(self.g.child(node, "conv"), prefix + "conv")

    def deep_factorized(self, node):                          # tfc NoisyDeepFactorized -> base DeepFactorized lists
        base = self.g.child(node, "_base", "base") if (self.g.has(node, "_base") or self.g.has(node, "base")) else node
        for kind, ours in (("_matrices", "matrix"), ("_biases", "bias"), ("_factors", "factor")):
            lst = self.g.child(base, kind, kind.lstrip("_"))
            i = 0
            while self.g.has(lst, str(i)):
                v = self.var(self.g.child(lst, str(i)))
                self.out[f"prior/{ours}_{i}"] = v.reshape(v.shape[0], v.shape[1]) if (ours != "matrix" and v.ndim == 3) else v
                i += 1


SIGNAL_STACKS = ("MBT2018Analysis", "MBT2018Synthesis", "BLS2017Analysis", "BLS2017Synthesis")


def _transform_shapes(cfg, cin):
    """Our parameter shapes of a registered transform (kernel sizes / channel counts of every layer)."""
    from .transforms import class_builder as transform_builder
    cfg = dict(cfg)
    t = transform_builder.build(cfg.pop("cls"), **cfg)
    return t._graph.shapes(cin)[0]


SYNTHESIS_MAPPERS = {"TwoLayerResSynthesis": "two_layer_res", "TwoLayerSynthesis": "two_layer", "JPEGLikeSynthesis": "jpeg_like"}


def load_reference_checkpoint(prefix, transform_config):
    """-> flat {name: ndarray} accepted by ``Model.set_weights`` for a checkpoint written by the reference's
    training loop (``tf.train.Checkpoint(model=model)``, common/train_lib.py:123-126) for the mean-scale
    hyperprior model with an ELIC / CNN analysis and a two-layer / JPEG-like synthesis."""
    tensors = read_bundle(prefix)
    if OBJECT_GRAPH_KEY not in tensors:
        raise KeyError(f"{prefix}: no {OBJECT_GRAPH_KEY}; not an object-based checkpoint")
    g = ObjectGraph(tensors[OBJECT_GRAPH_KEY])
    m = CheckpointMapper(tensors, g, prefix)
    model = g.child(0, "model")
    a_cls = transform_config["analysis"]["cls"]
    ana = g.child(model, "_analysis")
    if a_cls == "ElicAnalysis":
        m.elic_analysis(ana, "analysis/")
    elif a_cls == "CNNAnalysis":
        m.sequential_convs(ana, "analysis/", [f"layer_{i}" for i in range(4)])
    elif a_cls in SIGNAL_STACKS:
        m.signal_stack(ana, "analysis/", _transform_shapes(transform_config["analysis"], 3), "gdn_{}")
    else:
        raise NotImplementedError(f"checkpoint import for analysis {a_cls}")
    s_cls = transform_config["synthesis"]["cls"]
    if s_cls in SIGNAL_STACKS:
        bott = m.out[[k for k in m.out if k.startswith("analysis/") and k.endswith("/kernel")][-1]].shape[-1]
        m.signal_stack(g.child(model, "_synthesis"), "synthesis/", _transform_shapes(transform_config["synthesis"], bott), "igdn_{}")
    elif s_cls in SYNTHESIS_MAPPERS:
        getattr(m, SYNTHESIS_MAPPERS[s_cls])(g.child(model, "_synthesis"), "synthesis/")
    else:
        raise NotImplementedError(f"checkpoint import for synthesis {s_cls}")
    if not g.has(model, "_hyper_analysis"):                    # factorized prior (factorized/models.py): no hyper transforms
        m.deep_factorized(g.child(model, "_prior"))
        return m.out
    m.sequential_convs(g.child(model, "_hyper_analysis"), "hyper_analysis/", ["layer_0", "layer_1", "layer_2"])
    m.sequential_convs(g.child(model, "_hyper_synthesis"), "hyper_synthesis/", ["layer_0", "layer_1", "layer_2"])
    m.deep_factorized(g.child(model, "_prior"))
    return m.out


# ------------------------------------------------------------------------------------------ writing
class _GraphWriter:
    """Builds the TrackableObjectGraph + tensor dict of ``tf.train.Checkpoint(model=model)`` for the reference's
    classes -- the inverse of ``CheckpointMapper`` (same attribute names, same checkpoint keys)."""

    def __init__(self, weights):
        self.w = weights
        self.nodes = [({}, {})]
        self.tensors = OrderedDict()

    def add(self, parent, name):
        self.nodes.append(({}, {}))
        self.nodes[parent][0][name] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def var(self, parent, name, path, value):
        n = self.add(parent, name)
        key = f"{path}/{name}{VAR_SUFFIX}"
        self.nodes[n][1]["VARIABLE_VALUE"] = key
        self.tensors[key] = np.asarray(value)
        return n

    def conv(self, parent, name, path, ours):
        n = self.add(parent, name)
        self.var(n, "kernel", f"{path}/{name}", self.w[ours + "/kernel"])
        if ours + "/bias" in self.w:
            self.var(n, "bias", f"{path}/{name}", self.w[ours + "/bias"])
        return n

    def residual_block(self, parent, name, path, ours):
        blk = self.add(self.add(parent, name), "_block")
        for i in range(3):
            self.conv(blk, f"layer_with_weights-{i}", f"{path}/{name}/_block", f"{ours}/conv{i}")

    def elic_analysis(self, parent):
        seq = self.add(parent, "_transform")
        path = "model/_analysis/_transform"
        items = {k.split("/")[1] for k in self.w if k.startswith("analysis/")}
        nconv = sum(1 for i in items if i.startswith("conv"))
        nrb = sum(1 for i in items if i.startswith("rb"))
        per = nrb // (nconv - 1)                               # residual blocks after every conv but the last (elic.py:147-163)
        rbs = iter([f"rb{i}" for i in range(nrb)])
        take = lambda: [next(rbs) for _ in range(per)]
        names = (["conv0", *take()] if nconv == 4 else [])
        c = nconv - 3
        names += [f"conv{c}", *take(), "attn0", f"conv{c + 1}", *take(), f"conv{c + 2}", "attn1"]
        if set(names) != items:
            raise KeyError(f"analysis variables do not form an ElicAnalysis: {sorted(items)}")
        for i, item in enumerate(names):
            lw, ours = f"layer_with_weights-{i}", f"analysis/{item}"
            if item.startswith("conv"):
                self.conv(seq, lw, path, ours)
            elif item.startswith("rb"):
                self.residual_block(seq, lw, path, ours)
            else:
                att = self.add(seq, lw)
                trunk, branch = self.add(att, "_trunk"), self.add(att, "_attention_branch")
                for j in range(3):
                    self.residual_block(trunk, f"layer_with_weights-{j}", f"{path}/{lw}/_trunk", f"{ours}/trunk/rb{j}")
                    self.residual_block(branch, f"layer_with_weights-{j}", f"{path}/{lw}/_attention_branch", f"{ours}/branch/rb{j}")
                self.conv(branch, "layer_with_weights-3", f"{path}/{lw}/_attention_branch", f"{ours}/branch/conv")

    def signal_stack(self, parent, path, ours_prefix, gdn_names):
        i = 0
        while f"{ours_prefix}layer_{i}/kernel" in self.w:
            n = self.add(parent, f"layer_with_weights-{i}")
            lp = f"{path}/layer_with_weights-{i}"
            kp = self.add(n, "_kernel_parameter")
            _warn_rdft("export")
            self.var(kp, "rdft", f"{lp}/_kernel_parameter", kernel_to_rdft(self.w[f"{ours_prefix}layer_{i}/kernel"]))
            if f"{ours_prefix}layer_{i}/bias" in self.w:
                self.var(n, "_bias_parameter", lp, self.w[f"{ours_prefix}layer_{i}/bias"])
            g = ours_prefix + gdn_names.format(i)
            if g + "/beta" in self.w:
                act = self.add(n, "_activation")
                for pname, leaf in (("beta_parameter", "beta"), ("gamma_parameter", "gamma")):
                    p = self.add(act, pname)
                    self.var(p, "variable", f"{lp}/_activation/{pname}", gdn_parameter_variable(self.w[f"{g}/{leaf}"]))
            i += 1

    def sequential(self, parent, path, ours_prefix, count):
        for i in range(count):
            self.conv(parent, f"layer_with_weights-{i}", path, f"{ours_prefix}layer_{i}")

    def gdn1(self, parent, path, ours):
        act = self.add(parent, "activation")
        for pname, leaf in (("beta_parameter", "beta"), ("gamma_parameter", "gamma")):
            p = self.add(act, pname)
            self.var(p, "variable", f"{path}/activation/{pname}", gdn_parameter_variable(self.w[f"{ours}/{leaf}"]))


def save_reference_checkpoint(prefix, weights, transform_config, step=0):
    """Write ``weights`` (``Model.get_weights()`` naming) as the TensorBundle + object graph the reference's training
    loop produces (common/train_lib.py:123-126), so that the reference's ``eval.py`` (or ``load_reference_checkpoint``
    here) restores it.  Variables only: the Keras optimizer slots of the reference checkpoint are not written."""
    b = _GraphWriter(weights)
    model = b.add(0, "model")
    ana = b.add(model, "_analysis")
    a_cls = transform_config["analysis"]["cls"]
    if a_cls == "ElicAnalysis":
        b.elic_analysis(ana)
    elif a_cls == "CNNAnalysis":
        b.sequential(ana, "model/_analysis", "analysis/", 4)
    elif a_cls in SIGNAL_STACKS:
        b.signal_stack(ana, "model/_analysis", "analysis/", "gdn_{}")
    else:
        raise NotImplementedError(f"checkpoint export for analysis {a_cls}")
    syn = b.add(model, "_synthesis")
    s_cls = transform_config["synthesis"]["cls"]
    if s_cls == "TwoLayerResSynthesis":
        for n in ("base_conv", "res", "out_conv"):
            b.conv(syn, n, "model/_synthesis", f"synthesis/{n}")
        if "synthesis/act/beta" in weights:
            b.gdn1(syn, "model/_synthesis", "synthesis/act")
    elif s_cls == "TwoLayerSynthesis":
        c1 = b.conv(syn, "conv1", "model/_synthesis", "synthesis/conv1")
        b.conv(syn, "conv2", "model/_synthesis", "synthesis/conv2")
        if "synthesis/act/beta" in weights:
            b.gdn1(c1, "model/_synthesis/conv1", "synthesis/act")
    elif s_cls == "JPEGLikeSynthesis":
        b.conv(syn, "conv", "model/_synthesis", "synthesis/conv")
    elif s_cls in SIGNAL_STACKS:
        b.signal_stack(syn, "model/_synthesis", "synthesis/", "igdn_{}")
    else:
        raise NotImplementedError(f"checkpoint export for synthesis {s_cls}")
    if "hyper_analysis/layer_0/kernel" in weights:             # the factorized-prior model has no hyper transforms
        for tname in ("_hyper_analysis", "_hyper_synthesis"):
            b.sequential(b.add(model, tname), f"model/{tname}", f"{tname[1:]}/", 3)
    base = b.add(b.add(model, "_prior"), "_base")
    for kind, ours in (("_matrices", "matrix"), ("_biases", "bias"), ("_factors", "factor")):
        lst = b.add(base, kind)
        i = 0
        while f"prior/{ours}_{i}" in weights:
            v = np.asarray(weights[f"prior/{ours}_{i}"])
            b.var(lst, str(i), f"model/_prior/_base/{kind}", v if ours == "matrix" else v[..., None])
            i += 1
    tensors = OrderedDict(b.tensors)
    tensors[OBJECT_GRAPH_KEY] = ObjectGraph.serialize(b.nodes)
    tensors["save_counter" + VAR_SUFFIX] = np.array(int(step), np.int64)
    if a_cls in SIGNAL_STACKS or s_cls in SIGNAL_STACKS:
        tensors[WRITER_LAYOUT_KEY] = np.array(RDFT_LAYOUTS.index(RDFT_LAYOUT), np.int32)
    write_bundle(prefix, tensors)
    return prefix
