"""Packed checkpoints: 1 bit per binary weight + one fp32 alpha per output channel on disk.

The reference advertises the x32 storage saving of bit-packing (``README.md:22``) but stores fp32
``state_dict``s; its tests pin the state-dict schema (``test/test_binarize.py:95-110``: keys and values
survive a save/load of the converted model).  This module is the on-disk format either side of the
binary-convolution path:

* ``save_packed(model, path)`` writes, for every binary layer whose weight hook is an
  ``XNORWeightBinarizer``, ``sign(W)`` as bits (+ a zero mask only when some ``sign(W) == 0``) and the
  per-channel ``alpha``; every other tensor of the ``state_dict`` (BN statistics, the real-valued first
  and last layer, biases, learnable hook parameters) is stored verbatim.
* ``load_packed(path)`` returns a ``state_dict`` with the SAME keys, dtypes and shapes as the one that
  was saved — ``model.load_state_dict(load_packed(path))`` just works — whose binary weights are
  re-materialised so that the layer's forward is unchanged: ``XNORWeightBinarizer`` applied to the
  reconstructed tensor yields the same ``sign`` bits and (with the deterministic double-precision
  reduction of ``pack_weight``) bit-identical ``alpha``.  The latent fp32 magnitudes, which only matter
  for further training, are what is dropped.

Reconstruction.  Let ``s = sign(Wc)`` and ``alpha = mean|Wc|`` where ``Wc`` is ``W`` (or ``W`` minus its
mean over the input channels when ``center_weights``).
  no centring : ``W* = a * s``, ``a = alpha / mean|s|``   -> sign(W*) = s, mean|W*| = alpha
  centring    : ``W* = a * (s - mean_C(s))``, ``a = alpha / mean|s - mean_C(s)|``
                -> mean_C(W*) = 0, so centring is the identity on W*; sign and alpha as above
                (needs s != 0 and |mean_C(s)| < 1, i.e. not all signs equal along C; else kept in fp32)
  compute_alpha=False : ``alpha`` is not used by the forward; ``W* = s``.

Container: 8-byte magic ``BNNPACK1``, uint64 header length, UTF-8 JSON header, 64-byte aligned
little-endian blobs (same idea as safetensors; no pickle).
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

from .ops import XNORWeightBinarizer

MAGIC = b"BNNPACK1"
_ALIGN = 64


def _binary_weight_hooks(model: nn.Module) -> Dict[str, XNORWeightBinarizer]:
    """state_dict key of the weight -> its XNOR hook, for every binary layer of ``model``."""
    out = {}
    for name, mod in model.named_modules():
        hook = getattr(mod, "weight_pre_process", None)
        if isinstance(hook, XNORWeightBinarizer) and isinstance(getattr(mod, "weight", None), torch.Tensor):
            out[(name + "." if name else "") + "weight"] = hook
    return out


def _sign_alpha(w: torch.Tensor, center: bool) -> Tuple[np.ndarray, np.ndarray]:
    """``s`` (int8 in {-1,0,1}) and fp32 ``alpha[o]`` of ``ops.py:116-140``.  Centring is done with
    the very torch ops of the hook (so the stored signs are the ones the layer's forward sees); the
    alpha reduction runs in double, like ``pack_weight``."""
    w = w.detach().cpu().float()
    if center:
        w = w - w.mean(dim=1, keepdim=True)
    wc = w.numpy().astype(np.float64)
    s = np.sign(np.nan_to_num(wc, nan=0.0)).astype(np.int8)
    alpha = (np.abs(wc).reshape(wc.shape[0], -1).sum(axis=1) / np.prod(wc.shape[1:])).astype(np.float32)
    return s, alpha


def _reconstruct(s: np.ndarray, alpha: np.ndarray, center: bool, compute_alpha: bool) -> np.ndarray:
    sf = s.astype(np.float64)
    bshape = (-1,) + (1,) * (s.ndim - 1)
    if not compute_alpha:
        return sf.astype(np.float32)
    if not center:  # mean|a*s| = a * (non-zero fraction)  ->  a = alpha / mean|s|
        frac = np.abs(sf).reshape(s.shape[0], -1).mean(axis=1)
        a = np.divide(alpha.astype(np.float64), frac, out=np.zeros_like(frac), where=frac > 0)
        return (sf * a.reshape(bshape)).astype(np.float32)
    t = sf - sf.mean(axis=1, keepdims=True)
    scale = alpha.astype(np.float64) / np.abs(t).reshape(s.shape[0], -1).mean(axis=1)
    return (t * scale.reshape(bshape)).astype(np.float32)


def _representable(s: np.ndarray, center: bool) -> bool:
    if not center:
        return True
    # centred rows must keep their sign after re-centring: no exact zeros, not all signs equal along C
    return bool((s != 0).all() and (np.abs(s.astype(np.float64).mean(axis=1)) < 1.0).all())


def save_packed(model: nn.Module, path: str) -> Dict[str, int]:
    """Write ``model.state_dict()`` with binary weights bit-packed.  Returns byte counts
    ``{"file": ..., "fp32_state_dict": ..., "binary_weights_fp32": ..., "binary_weights_packed": ...}``."""
    hooks = _binary_weight_hooks(model)
    header = {"format": 1, "tensors": {}}
    blobs, offset = [], 0
    stats = {"fp32_state_dict": 0, "binary_weights_fp32": 0, "binary_weights_packed": 0}

    def add_blob(arr: np.ndarray) -> Dict[str, int]:
        nonlocal offset
        raw = np.ascontiguousarray(arr).tobytes()
        pad = (-offset) % _ALIGN
        blobs.append(b"\0" * pad + raw)
        ent = {"offset": offset + pad, "nbytes": len(raw)}
        offset += pad + len(raw)
        return ent

    for key, t in model.state_dict().items():
        a = t.detach().cpu().numpy()
        stats["fp32_state_dict"] += a.nbytes
        hook = hooks.get(key)
        packed = False
        if hook is not None and a.dtype == np.float32 and a.ndim in (2, 3, 4):
            s, alpha = _sign_alpha(t, hook.center_weights)
            if _representable(s, hook.center_weights):
                ent = {"kind": "xnor", "shape": list(a.shape), "center": bool(hook.center_weights),
                       "compute_alpha": bool(hook.compute_alpha),
                       "bits": add_blob(np.packbits((s > 0).reshape(-1), bitorder="little"))}
                if (s == 0).any():
                    ent["nz"] = add_blob(np.packbits((s != 0).reshape(-1), bitorder="little"))
                ent["alpha"] = add_blob(alpha.astype("<f4"))
                header["tensors"][key] = ent
                stats["binary_weights_fp32"] += a.nbytes
                stats["binary_weights_packed"] += sum(ent[k]["nbytes"] for k in ("bits", "nz", "alpha") if k in ent)
                packed = True
        if not packed:
            le = a.astype(a.dtype.newbyteorder("<"), copy=False)
            header["tensors"][key] = {"kind": "raw", "dtype": a.dtype.name, "shape": list(a.shape),
                                      "data": add_blob(le)}
    hjson = json.dumps(header).encode("utf-8")  # insertion order = state_dict order (kept on load)
    hjson += b" " * ((-(len(MAGIC) + 8 + len(hjson))) % _ALIGN)
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<Q", len(hjson)))
        fh.write(hjson)
        for b in blobs:
            fh.write(b)
    os.replace(tmp, path)
    stats["file"] = os.path.getsize(path)
    return stats


def _read(path: str):
    with open(path, "rb") as fh:
        if fh.read(len(MAGIC)) != MAGIC:
            raise ValueError(f"{path}: not a BNNPACK1 checkpoint")
        (hlen,) = struct.unpack("<Q", fh.read(8))
        header = json.loads(fh.read(hlen).decode("utf-8"))
        data = fh.read()
    if header.get("format") != 1:
        raise ValueError(f"{path}: unsupported format version {header.get('format')}")
    return header, data


def _blob(data: bytes, ent: Dict[str, int], dtype, expect_items: int = -1) -> np.ndarray:
    """One blob of the data section.  Header fields are untrusted input: offsets must lie inside the file and the
    byte count must be exactly what the tensor's shape and dtype imply."""
    off, nb = ent.get("offset"), ent.get("nbytes")
    if not isinstance(off, int) or not isinstance(nb, int) or off < 0 or nb < 0:
        raise ValueError("BNNPACK1: corrupt header (offset / nbytes)")
    if off + nb > len(data):
        raise ValueError("BNNPACK1: truncated file")
    item = np.dtype(dtype).itemsize
    if nb % item or (expect_items >= 0 and nb != expect_items * item):
        raise ValueError(f"BNNPACK1: blob of {nb} bytes does not match its tensor ({expect_items} x {item} bytes)")
    return np.frombuffer(data, dtype=dtype, count=ent["nbytes"] // np.dtype(dtype).itemsize, offset=ent["offset"])


def _shape_of(ent) -> Tuple[int, ...]:
    shape = ent.get("shape")
    if not isinstance(shape, list) or any(not isinstance(d, int) or d < 0 for d in shape):
        raise ValueError("BNNPACK1: corrupt header (shape)")
    return tuple(shape)


def _signs_from(header, data) -> Dict[str, Tuple[np.ndarray, np.ndarray, dict]]:
    out = {}
    for key, ent in header["tensors"].items():
        if ent["kind"] != "xnor":
            continue
        shape = _shape_of(ent)
        if len(shape) < 2:
            raise ValueError(f"BNNPACK1: binary weight {key!r} must have rank >= 2")
        n = int(np.prod(shape))
        nbits = (n + 7) // 8
        pos = np.unpackbits(_blob(data, ent["bits"], np.uint8, nbits), count=n, bitorder="little").astype(np.int8)
        s = 2 * pos - 1
        if "nz" in ent:
            s = s * np.unpackbits(_blob(data, ent["nz"], np.uint8, nbits), count=n, bitorder="little").astype(np.int8)
        out[key] = (s.reshape(shape), _blob(data, ent["alpha"], "<f4", shape[0]).astype(np.float32),
                    {"center": ent["center"], "compute_alpha": ent["compute_alpha"]})
    return out


def load_packed_signs(path: str) -> Dict[str, Tuple[np.ndarray, np.ndarray, dict]]:
    """The packed content itself: key -> (sign int8 array of the weight's shape, alpha fp32 [O], meta)."""
    return _signs_from(*_read(path))


def load_packed(path: str, map_location=None) -> Dict[str, torch.Tensor]:
    """``state_dict`` (same keys / dtypes / shapes as the one saved) with binary weights
    re-materialised as described in the module docstring."""
    header, data = _read(path)
    signs = _signs_from(header, data)
    sd = {}
    for key, ent in header["tensors"].items():
        if ent["kind"] == "xnor":
            s, alpha, meta = signs[key]
            arr = _reconstruct(s, alpha, meta["center"], meta["compute_alpha"])
        elif ent["kind"] == "raw":
            dt = np.dtype(ent["dtype"]).newbyteorder("<")
            shape = _shape_of(ent)
            arr = _blob(data, ent["data"], dt, int(np.prod(shape))).astype(np.dtype(ent["dtype"])).reshape(shape)
        else:
            raise ValueError(f"BNNPACK1: unknown tensor kind {ent['kind']!r}")
        t = torch.from_numpy(np.array(arr, copy=True))
        sd[key] = t.to(map_location) if map_location is not None else t
    return sd
