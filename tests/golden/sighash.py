"""Position-weighted checksums of sign tensors, identical for numpy / torch / packed bit planes.

Used to compare the DISCRETE state of a binary network between implementations: for every binary
convolution input x[N,C,H,W] the per-image value

    h[n] = sum_i  [x[n].flat[i] > 0] * w(i)   -   sum_i [x[n].flat[i] < 0] * w(i)         (int64)
    w(i) = ((i * 2654435761 + 12345) mod 2^32) >> 8                                       (24 bits)

changes whenever a single sign() result changes (and for a single flip |dh| identifies w(i), hence
the position).  Two implementations with equal h for every layer of an image went through the same
integers everywhere; what is left between their logits is plain floating-point rounding.
"""
from __future__ import annotations

import numpy as np


def weights_np(n: int) -> np.ndarray:
    i = np.arange(n, dtype=np.uint64)
    return (((i * np.uint64(2654435761) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)) >> np.uint64(8)).astype(np.int64)


def sign_hash_np(x: np.ndarray) -> np.ndarray:
    """x: float [N, ...] -> int64 [N]."""
    n = x.shape[0]
    flat = x.reshape(n, -1)
    w = weights_np(flat.shape[1])
    return ((flat > 0).astype(np.int64) * w).sum(1) - ((flat < 0).astype(np.int64) * w).sum(1)


def sign_hash_torch(x):
    """x: float tensor [N, ...] (any device) -> int64 tensor [N] on that device."""
    import torch
    n = x.shape[0]
    flat = x.reshape(n, -1)
    w = torch.from_numpy(weights_np(flat.shape[1])).to(x.device)
    return ((flat > 0).to(torch.int64) * w).sum(1) - ((flat < 0).to(torch.int64) * w).sum(1)


def sign_hash_planes(P, M, C: int):
    """Bit planes P, M: int64 tensors [N, ceil(C/64), H, W] (format of include/bnn_hip.h) -> int64 [N]."""
    import torch
    n, g, h, wd = P.shape
    w = torch.from_numpy(weights_np(g * 64 * h * wd)).to(P.device).view(g, 64, h, wd)[: , :, :, :]
    out = torch.zeros(n, dtype=torch.int64, device=P.device)
    shifts = torch.arange(64, device=P.device, dtype=torch.int64).view(1, 1, 64, 1, 1)
    for plane, sgn in ((P, 1), (M, -1)):
        for n0 in range(0, n, 16):   # bound the [n,g,64,h,w] temporary
            bits = (plane[n0:n0 + 16].unsqueeze(2) >> shifts) & 1          # [n,g,64,h,w]; channel = 64*g + b
            out[n0:n0 + 16] += sgn * (bits * w.view(1, g, 64, h, wd)).sum((1, 2, 3, 4))
    # channels >= C are pad bits (always 0): their weights never count
    return out
