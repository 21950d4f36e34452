"""Pure-Python restatement of the interleaved rANS stream format of csrc/rans.hip (TEST INFRASTRUCTURE; small cases).

Not a reference algorithm: the reference has no bitstream (compression=False, mshyper/models.py:246-251).  This pins
the product's own wire format: LANES (8 .. 64) lanes, each a 32-bit state with 16-bit renormalisation and 16-bit probability
precision, sharing one sequence of 16-bit words; element 64 j + l of the segment belongs to lane l at step j;
ESCAPE = last symbol of a table, followed by (value + 32768) as a raw 16-bit word;
stream = [lane 0 state hi, lo, lane 1 state hi, lo, ...] [words in decode order]."""
from __future__ import annotations

import math

M = 1 << 16
LANES = 64
NUM_SCALES, SCALE_MIN, SCALE_MAX = 64, 0.11, 256.0       # mshyper/models.py:28-32


def _ndtr(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def quantize_pmf(pmf, escape_mass):
    """Frequencies at 16-bit precision: 1 + floor(p (65536 - n)) each, the remainder to the most probable symbol;
    the last entry is ESCAPE."""
    p = [max(float(v), 0.0) for v in pmf] + [max(float(escape_mass), 0.0)]
    tot = sum(p)
    p = [v / tot for v in p]
    n = len(p)
    f = [1 + int(math.floor(v * (M - n))) for v in p]
    f[max(range(n), key=lambda i: p[i])] += M - sum(f)
    assert min(f) >= 1 and sum(f) == M
    return f


def normal_tables(min_pmf=2.0 ** -17, max_half_width=4095):
    """One table per integer scale index k: sigma_k = SCALE_FN(k), symbols |v| <= L_k with pmf >= 2^-17, then ESCAPE."""
    factor = (math.log(SCALE_MAX) - math.log(SCALE_MIN)) / (NUM_SCALES - 1.0)
    tabs = []
    for k in range(NUM_SCALES):
        sigma = math.exp(math.log(SCALE_MIN) + factor * k)

        def pmf_at(t):
            if t <= 0:
                return _ndtr((t + 0.5) / sigma) - _ndtr((t - 0.5) / sigma)
            return _ndtr(-(t - 0.5) / sigma) - _ndtr(-(t + 0.5) / sigma)

        L = 0
        while L < max_half_width and pmf_at(L + 1) >= min_pmf:
            L += 1
        tabs.append((-L, quantize_pmf([pmf_at(t) for t in range(-L, L + 1)], 2.0 * _ndtr(-(L + 0.5) / sigma))))
    return tabs


def _cdf(f):
    c = [0]
    for fi in f:
        c.append(c[-1] + int(fi))
    return c


def encode_stream(values, tids, tables, LANES=LANES):
    """tables[t] = (vmin, freqs incl. ESCAPE last).  Returns the list of uint16 words in stream order.
    Works backward exactly as the kernel does, so ``rev`` collects words from the last address to the first."""
    values, tids = [int(v) for v in values], [int(t) for t in tids]
    n = len(values)
    steps = -(-n // LANES)
    x = [M] * LANES
    rev = []
    for j in reversed(range(steps)):
        live = [l for l in range(LANES) if j * LANES + l < n]
        sym = {}
        for l in reversed(live):                      # ESCAPE payloads first: highest lane at the highest address
            v = values[j * LANES + l]
            vmin, f = tables[tids[j * LANES + l]]
            s = v - vmin
            if s < 0 or s >= len(f) - 1:
                rev.append(x[l] & 0xFFFF)
                x[l] = (x[l] & 0xFFFF0000) | (min(max(v, -32768), 32767) + 32768)
                s = len(f) - 1
            sym[l] = s
        for l in reversed(live):
            vmin, f = tables[tids[j * LANES + l]]
            c = _cdf(f)
            fr, cl = c[sym[l] + 1] - c[sym[l]], c[sym[l]]
            if x[l] >= (fr << 16):
                rev.append(x[l] & 0xFFFF)
                x[l] >>= 16
            x[l] = ((x[l] // fr) << 16) + (x[l] % fr) + cl
    for l in reversed(range(LANES)):
        rev.append(x[l] & 0xFFFF)
        rev.append(x[l] >> 16)
    return list(reversed(rev))


def decode_stream(words, tids, tables, LANES=LANES):
    tids = [int(t) for t in tids]
    n = len(tids)
    steps = -(-n // LANES)
    x = [(words[2 * l] << 16) | words[2 * l + 1] for l in range(LANES)]
    pos, out = 2 * LANES, [0] * n
    for j in range(steps):
        live = [l for l in range(LANES) if j * LANES + l < n]
        esc = []
        for l in live:                                 # phase 1: every lane decodes its symbol from its own state
            vmin, f = tables[tids[j * LANES + l]]
            c = _cdf(f)
            slot = x[l] & 0xFFFF
            s = max(i for i in range(len(f)) if c[i] <= slot)
            x[l] = (c[s + 1] - c[s]) * (x[l] >> 16) + slot - c[s]
            out[j * LANES + l] = s + vmin
            if s == len(f) - 1:
                esc.append(l)
        for l in live:                                 # phase 2: refills in lane order
            if x[l] < M:
                x[l] = (x[l] << 16) | words[pos]
                pos += 1
        for l in esc:                                  # phase 3: escaped values, then their refills in lane order
            out[j * LANES + l] = (x[l] & 0xFFFF) - 32768
            x[l] >>= 16
        for l in esc:
            x[l] = (x[l] << 16) | words[pos]
            pos += 1
    assert all(v == M for v in x) and pos == len(words), "stream did not terminate at its initial state"
    return out
