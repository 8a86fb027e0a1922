"""Hook-parameter transfer used by ``from_module(..., update=True)`` (``bnn/layers/helpers.py``)."""
from __future__ import annotations

from dataclasses import fields

import torch


def copy_paramters(source_mod: torch.nn.Module, target_mod: torch.nn.Module, bconfig) -> None:
    """Copy same-shaped hook parameters (e.g. a learned ``alpha``) from ``source_mod``.

    The (misspelled) name is the reference's.  Unlike ``bnn/layers/helpers.py:7-17`` a missing
    hook on either side is skipped instead of raising ``AttributeError``.
    """
    for f in fields(bconfig):
        src = getattr(source_mod, f.name, None)
        dst = getattr(target_mod, f.name, None)
        if src is None or dst is None:
            continue
        dst_params = dict(dst.named_parameters())
        for name, p in src.named_parameters():
            q = dst_params.get(name)
            if q is not None and q.shape == p.shape:
                q.data.copy_(p.data)
