"""Portable, torch-independent deterministic tensor generator for tests and fixtures.

Every value is a pure function of (seed, flat index): a splitmix64 hash -> 4 uniforms in
[0,1) on a 2^-24 grid -> Irwin-Hall "normal" (sum of four uniforms, centred and scaled to unit
variance).  Only integer arithmetic and exactly-representable float64 additions are used, so the
GPU box regenerates bit-identical inputs without the reference and without any RNG coupling.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def seed_of(*parts) -> int:
    """Stable 32-bit seed from strings / ints (crc32 of the repr)."""
    return zlib.crc32("|".join(str(p) for p in parts).encode()) & 0x7FFFFFFF


def uniform(seed: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(idx * np.uint64(4) + (np.uint64(seed) << np.uint64(34)))
    return ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24)).reshape(shape)


def normal(seed: int, shape) -> np.ndarray:
    """~N(0,1) (Irwin-Hall, n=4): float32 array."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    acc = np.zeros(n, np.float64)
    with np.errstate(over="ignore"):
        base = idx * np.uint64(4) + (np.uint64(seed) << np.uint64(34))
        for k in range(4):
            h = _splitmix64(base + np.uint64(k))
            acc += (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    # + 2^-25: half a grid step, so that no value is exactly 0.0 (zeros are injected explicitly
    # by the "sparse"/"relu"/"withzeros" kinds, never by accident)
    return ((acc - 2.0 + 2.0 ** -25) * np.sqrt(3.0)).astype(np.float32).reshape(shape)


def activation(kind: str, seed: int, shape) -> np.ndarray:
    """Synthetic layer inputs covering the sign()/zero semantics of the reference.

    normal   : N(0,1)                       (no exact zeros -> pure XNOR regime)
    relu     : relu(N(0,1))                 (~50 % exact zeros, no negatives: what ResNet feeds)
    negrelu  : -relu(N(0,1))
    sparse   : N(0,1) with 30 % exact zeros and whole zero rows/channels
    special  : N(0,1) sprinkled with NaN, +-inf, +-0.0, +-denormals
    """
    x = normal(seed, shape)
    if kind == "normal":
        return x
    if kind == "relu":
        return np.maximum(x, 0).astype(np.float32)
    if kind == "negrelu":
        return (-np.maximum(x, 0)).astype(np.float32)
    if kind == "sparse":
        u = uniform(seed + 1, shape)
        x = np.where(u < 0.3, 0.0, x).astype(np.float32)
        if x.ndim == 4 and x.shape[2] > 1:
            x[:, 0] = 0.0           # a dead channel
            x[:, :, 0, :] = 0.0     # a zero row
        return x
    if kind == "special":
        u = uniform(seed + 2, shape)
        x = x.copy()
        x[u < 0.02] = np.nan
        x[(u >= 0.02) & (u < 0.04)] = np.inf
        x[(u >= 0.04) & (u < 0.06)] = -np.inf
        x[(u >= 0.06) & (u < 0.10)] = 0.0
        x[(u >= 0.10) & (u < 0.14)] = -0.0
        x[(u >= 0.14) & (u < 0.17)] = np.float32(1e-45)
        x[(u >= 0.17) & (u < 0.20)] = np.float32(-1e-45)
        x[(u >= 0.20) & (u < 0.22)] = np.float32(1e-39)
        x[(u >= 0.22) & (u < 0.24)] = np.float32(-1e-39)
        return x.astype(np.float32)
    raise ValueError(kind)


def conv_weight(kind: str, seed: int, shape) -> np.ndarray:
    """kaiming: N(0, sqrt(2/fan_out)) as bnn/models/resnet.py:105; default: U(-b,b), b=1/sqrt(fan_in)."""
    fan_in = int(np.prod(shape[1:]))
    fan_out = int(shape[0] * np.prod(shape[2:])) if len(shape) > 2 else shape[0]
    if kind == "kaiming":
        return (normal(seed, shape) * np.sqrt(2.0 / fan_out)).astype(np.float32)
    if kind == "default":
        b = 1.0 / np.sqrt(fan_in)
        return ((uniform(seed, shape) * 2 - 1) * b).astype(np.float32)
    if kind == "withzeros":  # pruned weights: sign(0) == 0 must be honoured
        w = (normal(seed, shape) * np.sqrt(2.0 / fan_out)).astype(np.float32)
        w[uniform(seed + 7, shape) < 0.2] = 0.0
        return w
    raise ValueError(kind)


def model_state(state_shapes: dict, seed: int) -> dict:
    """Deterministic parameters/buffers for a whole network, keyed like ``state_dict()``.

    conv/linear weights: kaiming-normal fan_out; BN: gamma~U(.5,1.5), beta~N(0,.3),
    running_mean~N(0,.5), running_var~U(.5,1.5)  (default BN statistics make a random-init
    binary net nearly degenerate); biases ~N(0,.1).
    """
    out = {}
    for name, shape in state_shapes.items():
        s = seed_of(seed, name)
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, np.int64)
        elif name.endswith("running_var"):
            out[name] = (0.5 + uniform(s, shape)).astype(np.float32)
        elif name.endswith("running_mean"):
            out[name] = (0.5 * normal(s, shape)).astype(np.float32)
        elif name.endswith("alpha"):  # BasicScaleBinarizer scale, [1,C,1,1]
            out[name] = (0.5 + uniform(s, shape)).astype(np.float32)
        elif len(shape) >= 2:
            out[name] = conv_weight("kaiming", s, shape)
        elif name.endswith("weight"):  # 1-D weight: a norm layer's gamma (or PReLU slope)
            out[name] = (0.5 + uniform(s, shape)).astype(np.float32)
        else:  # biases / beta
            out[name] = (0.3 * normal(s, shape)).astype(np.float32)
    return out
