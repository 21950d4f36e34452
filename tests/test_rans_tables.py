"""The rANS decoder's own view of the entropy tables (entropy_coding.decoder_entries / start_tables; include/sntc.h,
sntc_rans_decode's dec / lut arguments), on the CPU: a numpy emulation of csrc/rans.hip::rans_decode_fast_kernel's symbol
search over EVERY slot of every table must find the symbol a plain search of the cdf finds, and recover its (start, frequency)
from the packed entry -- including the tables where the packing's edge cases live (a frequency of 65535, a last symbol
starting at 65535, single-count tails)."""
import numpy as np
import pytest

from shallow_ntc_amd import entropy_coding as ec


def emulate_fast_search(tabs, bits):
    cdfs = [np.concatenate([[0], np.cumsum(f)[:-1]]).astype(np.uint16) for _, f in tabs]
    lut, lmeta = ec.start_tables(cdfs, bits)
    dec = ec.decoder_entries(tabs).astype(np.uint64)
    assert len(dec) % 4 == 0 and len(lut) % 8 == 0
    slots = np.arange(65536, dtype=np.uint64)
    key = (slots << np.uint64(16)) | np.uint64(0xFFFE)
    off = 0
    worst = 0
    for t, ((_, f), cdf, b) in enumerate(zip(tabs, cdfs, bits)):
        n = len(f)
        base = off + 3 * t                                   # meta.x + 3 t, as the kernel builds it
        assert (dec[base + n:base + n + 3] == 0xFFFFFFFF).all()
        assert lmeta[t] & 31 == b
        lo = lut[(int(lmeta[t]) >> 5) + (slots >> np.uint64(16 - b)).astype(np.int64)].astype(np.int64)
        esel = np.zeros(65536, np.uint64)
        rounds = 0
        while True:
            rounds += 1
            c = [dec[base + lo + k] for k in range(4)]       # lo + 3 <= n + 2: inside the sentinels
            g1, g2, g3 = key >= c[1], key >= c[2], key >= c[3]
            esel = np.where(g2, c[2], np.where(g1, c[1], c[0]))
            lo = lo + g1.astype(np.int64) + g2.astype(np.int64)
            if not g3.any():
                break
            lo = lo + g3.astype(np.int64)
            assert rounds < 20000
        worst = max(worst, rounds)
        want = np.searchsorted(cdf.astype(np.int64), slots.astype(np.int64), side="right") - 1
        np.testing.assert_array_equal(lo, want)
        np.testing.assert_array_equal((esel >> np.uint64(16)).astype(np.int64), cdf.astype(np.int64)[want])
        np.testing.assert_array_equal((esel & np.uint64(0xFFFF)).astype(np.int64) + 1, np.asarray(f, np.int64)[want])
        off += n
    return worst


def test_every_slot_of_the_scale_tables():
    tabs = ec.normal_tables()
    for extra in (0, 1, -2):
        bits = [min(16, max(0, int(np.ceil(np.log2(len(f)))) + extra)) for _, f in tabs]
        worst = emulate_fast_search(tabs, bits)
        assert worst <= (24 if extra >= 0 else 96), worst      # thin tails: a bucket of up to 2^(16 - bits) single-count symbols, three per round


def test_packing_edge_cases():
    rng = np.random.default_rng(3)
    tabs = [(0, np.array([65535, 1])), (0, np.array([1, 65535])), (-1, np.array([1, 65534, 1])), (0, np.full(256, 256)),
            (-3, np.array([1] * 100 + [65536 - 199] + [1] * 99)), (0, np.array([32768, 32767, 1]))]
    f = rng.integers(1, 40, 3000)
    f[1500] += 65536 - f.sum()
    tabs.append((-1500, f))
    for bits in ([0] * len(tabs), [1, 1, 2, 8, 8, 2, 12], [16] * len(tabs), [3] * len(tabs)):
        emulate_fast_search(tabs, bits)
    with pytest.raises(AssertionError):
        ec.decoder_entries([(0, np.array([65536]))])          # a one-symbol table has no ESCAPE and a frequency that does not pack
