#!/usr/bin/env python3
"""The polynomial behind csrc/entropy.hip::erfcx_pos (no GPU): f(q) = (1 + 2x) exp(x^2) erfc(x) with q = (x - 2) / (x + 2) maps
[0, inf) to [-1, 1) and is smooth up to q = 1 (f -> 2 / sqrt(pi)); Chebyshev interpolation of degree 12, printed as monomial
float32 coefficients (highest first, Horner order), then checked the way the kernel evaluates it (float32 fma Horner) against
SciPy's float64 erfcx on a dense grid, and the whole -log2 P formula of normal_bits_fast against the float64 definition on
2e6 synthetic latents (SURVEY.md 8d entropy set).  python tools/fit_erfcx.py"""
import math

import numpy as np
import scipy.special as sp
from numpy.polynomial import chebyshev as Cb

f32 = np.float32
DEG = 12


def f_of_q(q):
    q = np.asarray(q, float)
    out = np.full_like(q, 2 / math.sqrt(math.pi))
    m = q < 1 - 1e-12
    x = 2 * (1 + q[m]) / (1 - q[m])
    out[m] = (1 + 2 * x) * sp.erfcx(x)
    return out


mono = Cb.cheb2poly(Cb.chebinterpolate(f_of_q, DEG)).astype(f32)
print("coefficients, highest degree first:")
for c in mono[::-1]:
    print(f"  {float(c)!r}f")


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(f32)


def erfcx32(x):
    q = ((x - f32(2)) / (x + f32(2))).astype(f32)
    p = np.full_like(q, mono[-1])
    for k in range(DEG - 1, -1, -1):
        p = fma(p, q, mono[k])
    return (p / (f32(1) + f32(2) * x)).astype(f32)


x = np.concatenate([np.linspace(0, 10, 200001), np.logspace(1, 4, 20001)]).astype(f32)
rel = np.abs(erfcx32(x).astype(np.float64) - sp.erfcx(x.astype(np.float64))) / sp.erfcx(x.astype(np.float64))
print(f"erfcx, float32 Horner vs float64: max relative error {rel.max():.3e}, mean {rel.mean():.3e}")

LN_MIN, FAC = f32(-2.2072749131897207), f32(0.12305479932808384)
exp32 = lambda v: np.exp(v.astype(np.float64)).astype(f32)
log32 = lambda v: np.log(v.astype(np.float64)).astype(f32)


def bits_fast(y, mu, raw):
    idx = np.minimum(exp32(raw), f32(63))
    isg = exp32(-fma(np.full_like(idx, FAC), idx, LN_MIN))
    a = np.abs(np.rint((y - mu).astype(f32)))
    zl, zu = ((a - f32(.5)) * isg).astype(f32), ((a + f32(.5)) * isg).astype(f32)
    s = f32(0.70710678118654752)
    cu, cl = erfcx32((zu * s).astype(f32)), erfcx32((np.abs(zl) * s).astype(f32))
    nz = a >= 1
    t = exp32(-np.where(nz, ((a * isg).astype(f32) * isg).astype(f32), ((f32(.5) * zu).astype(f32) * zu).astype(f32)))
    e = (t * cu).astype(f32)
    ser = e.astype(np.float64)
    ser = (-ser * (1 + ser * (.5 + ser * (1 / 3 + ser * (.25 + ser * .2))))).astype(f32)
    d = np.where(nz, (f32(.5) * (cl - e)).astype(f32), (f32(1) - e).astype(f32))
    lnp = log32(np.maximum(d, f32(1e-45)))
    lnp = np.where(~nz & (e < f32(0.0625)), ser, lnp)
    lnp = np.where(nz, fma((f32(-.5) * zl).astype(f32), zl, 0) + lnp, lnp).astype(f32)
    return -(lnp.astype(np.float64) * 1.4426950408889634).astype(f32)          # the kernel carries 1 / ln 2 in two terms


def bits_f64(y, mu, raw):
    idx = np.clip(np.exp(raw.astype(np.float64)), 0, 63)
    sg = np.exp(math.log(.11) + 0.12305479932808384 * idx)
    v = np.rint((y - mu).astype(f32)).astype(np.float64)
    hi, lo = (v + .5) / sg, (v - .5) / sg
    c = sp.log_ndtr(-hi) < sp.log_ndtr(hi)
    big = np.where(c, sp.log_ndtr(-lo), sp.log_ndtr(hi))
    small = np.where(c, sp.log_ndtr(-hi), sp.log_ndtr(lo))
    return -(big + np.log1p(-np.exp(small - big))) / math.log(2)


rng = np.random.default_rng(0)
n = 2_000_000
mu = rng.standard_normal(n).astype(f32)
for name, raw, y in (("SURVEY 8d set (raw ~ U(-3, 4.3), Laplace(0, 2))", rng.uniform(-3, 4.3, n).astype(f32), (mu + rng.laplace(0, 2, n)).astype(f32)),
                     ("narrow set (raw ~ U(-6, 1.5), Laplace(0, .4))", rng.uniform(-6, 1.5, n).astype(f32), (mu + rng.laplace(0, .4, n)).astype(f32))):
    got, ref = bits_fast(y, mu, raw), bits_f64(y, mu, raw)
    err = got.astype(np.float64) - ref
    print(f"{name}: sum of bits {ref.sum():.1f}, relative error of the sum {err.sum() / ref.sum():+.2e}, mean |error| per symbol {np.abs(err).mean():.2e} bits")
