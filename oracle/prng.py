"""TEST INFRASTRUCTURE — deterministic, libm-free synthetic weights / inputs.

Both the golden generator (this container, next to the imported reference) and the
GPU-box parity tests regenerate the SAME 31,030,658 parameters from this counter
based generator, so only outputs need to be committed as fixtures.
Only integer ops and exact float64 adds/multiplies are used (no log/cos), so the
stream is bit-reproducible across machines.

Layer table restates the reference's declaration order and init std formulas
(network.py:23-58, :70-105; SURVEY quirk Q1: std = 2 / sqrt(N), conv11c sqrt(2)).
"""
from collections import OrderedDict
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def _u24(seed, stream, idx, draw):
    """uniform in [0,1) with 24 bits, as float64 (exact)."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        ctr = idx * np.uint64(4) + np.uint64(draw)
        z = _splitmix64(ctr ^ key)
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def uniform01(seed, stream, n):
    idx = np.arange(n, dtype=np.uint64)
    return _u24(seed, stream, idx, 0)


def pseudo_normal(seed, stream, n):
    """Zero-mean unit-variance (Irwin-Hall of 4 uniforms; exact arithmetic)."""
    idx = np.arange(n, dtype=np.uint64)
    s = _u24(seed, stream, idx, 0) + _u24(seed, stream, idx, 1) \
        + _u24(seed, stream, idx, 2) + _u24(seed, stream, idx, 3)
    return (s - 2.0) * 1.7320508075688772  # var of sum = 4/12


def layer_table(base=64):
    """[(name, kind, cin, cout, k, std)] in the reference's declaration order."""
    c = [base, base * 2, base * 4, base * 8, base * 16]
    t = []

    def conv(name, ci, co, n_in=None, k=3):
        n_in = ci * 9 if n_in is None else n_in
        t.append((name, "conv", ci, co, k, 2.0 / (n_in ** 0.5)))

    def up(name, ci, co):
        t.append((name, "upconv", ci, co, 2, 2.0 / ((ci * 9) ** 0.5)))

    t.append(("conv11c", "conv", 1, c[0], 3, 2 ** 0.5))
    conv("conv12c", c[0], c[0])
    conv("conv21c", c[0], c[1]); conv("conv22c", c[1], c[1])
    conv("conv31c", c[1], c[2]); conv("conv32c", c[2], c[2])
    conv("conv41c", c[2], c[3]); conv("conv42c", c[3], c[3])
    conv("conv51c", c[3], c[4]); conv("conv52c", c[4], c[4])
    up("upconv4", c[4], c[3])
    conv("conv41e", c[4], c[3], n_in=c[3] * 9 + c[3] * 4); conv("conv42e", c[3], c[3])
    up("upconv3", c[3], c[2])
    conv("conv31e", c[3], c[2], n_in=c[2] * 9 + c[2] * 4); conv("conv32e", c[2], c[2])
    up("upconv2", c[2], c[1])
    conv("conv21e", c[2], c[1], n_in=c[1] * 9 + c[1] * 4); conv("conv22e", c[1], c[1])
    up("upconv1", c[1], c[0])
    conv("conv11e", c[1], c[0], n_in=c[0] * 9 + c[0] * 4); conv("conv12e", c[0], c[0])
    t.append(("finalconv", "conv", c[0], 2, 1, 2.0 / ((c[0] * 9) ** 0.5)))
    return t


def make_params(seed=0, base=64, dtype=np.float32):
    """OrderedDict of the 46 state-dict tensors (weight, bias per layer)."""
    out = OrderedDict()
    for li, (name, kind, ci, co, k, std) in enumerate(layer_table(base)):
        shape = (co, ci, k, k) if kind == "conv" else (ci, co, k, k)
        n = int(np.prod(shape))
        w = pseudo_normal(seed, 2 * li, n) * std
        out[name + ".weight"] = w.reshape(shape).astype(dtype)
        # PyTorch default bias init U(-1/sqrt(fan_in), +) ; fan_in = shape[1]*k*k
        bound = 1.0 / ((shape[1] * k * k) ** 0.5)
        b = (uniform01(seed, 2 * li + 1, co) * 2.0 - 1.0) * bound
        out[name + ".bias"] = b.astype(dtype)
    return out


def make_input(seed, B, S, dtype=np.float32):
    """U[0,1) images [B,1,S,S] (the reference normalises to [0,1], data.py:134,188)."""
    return uniform01(seed, 1000, B * S * S).reshape(B, 1, S, S).astype(dtype)


def make_labels(seed, B, So):
    """Bernoulli(0.5) int64 labels [B,1,So,So]."""
    return (uniform01(seed, 1001, B * So * So) >= 0.5).astype(np.int64).reshape(B, 1, So, So)


def make_cotangent(seed, shape, dtype=np.float32):
    n = int(np.prod(shape))
    return (pseudo_normal(seed, 1002, n) * 1e-3).reshape(shape).astype(dtype)
