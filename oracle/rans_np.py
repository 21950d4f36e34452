"""Pure-Python restatement of the rANS stream format of csrc/rans.hip (TEST INFRASTRUCTURE; small cases only).

Not a reference algorithm: the reference has no bitstream (compression=False, mshyper/models.py:246-251).  This
pins the product's own wire format: 32-bit state, 16-bit words, 16-bit probability precision, ESCAPE = last
symbol followed by (value + 32768) as a uniform 16-bit symbol; stream = [state hi, state lo, words ...]."""
from __future__ import annotations

M = 1 << 16


def _put(x, f, c, words):
    if x >= (f << 16):
        words.append(x & 0xFFFF)
        x >>= 16
    return ((x // f) << 16) + (x % f) + c, words


def encode_stream(values, tids, tables):
    """tables[t] = (vmin, freqs incl. ESCAPE last).  Returns the list of uint16 words in stream order."""
    x, words = M, []
    for v, t in zip(reversed(list(values)), reversed(list(tids))):
        vmin, f = tables[t]
        cdf = [0]
        for fi in f:
            cdf.append(cdf[-1] + int(fi))
        sym = int(v) - vmin
        if sym < 0 or sym >= len(f) - 1:
            x, words = _put(x, 1, min(max(int(v), -32768), 32767) + 32768, words)
            sym = len(f) - 1
        x, words = _put(x, cdf[sym + 1] - cdf[sym], cdf[sym], words)
    words.append(x & 0xFFFF)
    words.append(x >> 16)
    return list(reversed(words))


def decode_stream(words, tids, tables):
    x = (words[0] << 16) | words[1]
    pos, out = 2, []
    for t in tids:
        vmin, f = tables[t]
        cdf = [0]
        for fi in f:
            cdf.append(cdf[-1] + int(fi))
        slot = x & 0xFFFF
        sym = max(i for i in range(len(f)) if cdf[i] <= slot)
        x = (cdf[sym + 1] - cdf[sym]) * (x >> 16) + slot - cdf[sym]
        if x < M:
            x = (x << 16) | words[pos]
            pos += 1
        v = sym + vmin
        if sym == len(f) - 1:
            v = (x & 0xFFFF) - 32768
            x >>= 16
            if x < M:
                x = (x << 16) | words[pos]
                pos += 1
        out.append(v)
    assert x == M and pos == len(words), "stream did not terminate at its initial state"
    return out
