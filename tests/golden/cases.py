"""Shared definition of the golden cases (used by make_golden.py here and by the tests)."""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Optional, Tuple

import numpy as np

from . import gen


@dataclass(frozen=True)
class LayerCase:
    name: str
    N: int
    C: int
    H: int
    W: int
    O: int
    k: int
    stride: int = 1
    pad: int = 0
    act: str = "normal"        # gen.activation kind
    winit: str = "kaiming"     # gen.conv_weight kind
    center: bool = False
    compute_alpha: bool = True
    bias: bool = False
    post: str = "identity"     # identity | scale
    dilation: int = 1

    @property
    def xshape(self) -> Tuple[int, int, int, int]:
        return (self.N, self.C, self.H, self.W)

    @property
    def wshape(self) -> Tuple[int, int, int, int]:
        return (self.O, self.C, self.k, self.k)

    def tensors(self):
        s = gen.seed_of("layer", self.name)
        x = gen.activation(self.act, s, self.xshape)
        w = gen.conv_weight(self.winit, s + 11, self.wshape)
        b = (0.1 * gen.normal(s + 12, (self.O,))).astype(np.float32) if self.bias else None
        sc = (0.5 + gen.uniform(s + 13, (self.O,))).astype(np.float32) if self.post == "scale" else None
        return x, w, b, sc


# every distinct binary-conv shape of ResNet-18 at reduced batch / spatial size,
# plus tails (C not a multiple of 64), strides, 1x1, bias/scale, centring, zero semantics.
LAYER_CASES = [
    # R18 shapes (channels/kernel/stride/pad exact; spatial reduced where noted)
    LayerCase("l1_64x56", 1, 64, 56, 56, 64, 3, 1, 1, act="relu"),
    LayerCase("l2_0_c1_s2", 1, 64, 28, 28, 128, 3, 2, 1, act="relu"),
    LayerCase("l2_128x28", 1, 128, 28, 28, 128, 3, 1, 1, act="relu"),
    LayerCase("l2_ds_1x1", 2, 64, 14, 14, 128, 1, 1, 0, act="relu"),
    LayerCase("l3_0_c1_s2", 1, 128, 14, 14, 256, 3, 2, 1, act="relu"),
    LayerCase("l3_256x14", 1, 256, 14, 14, 256, 3, 1, 1, act="relu"),
    LayerCase("l3_ds_1x1", 2, 128, 7, 7, 256, 1, 1, 0, act="relu"),
    LayerCase("l4_0_c1_s2", 1, 256, 14, 14, 512, 3, 2, 1, act="relu"),
    LayerCase("l4_512x7", 2, 512, 7, 7, 512, 3, 1, 1, act="relu"),
    LayerCase("l4_ds_1x1", 2, 256, 7, 7, 512, 1, 1, 0, act="relu"),
    # BASELINE config 2 shape at reduced batch/spatial: 128->128 3x3 p1
    LayerCase("c2_normal", 2, 128, 12, 12, 128, 3, 1, 1, act="normal"),
    LayerCase("c2_relu", 2, 128, 12, 12, 128, 3, 1, 1, act="relu"),
    # sign()/zero semantics
    LayerCase("neg_only", 2, 64, 9, 9, 32, 3, 1, 1, act="negrelu"),
    LayerCase("sparse_rows", 2, 64, 9, 9, 32, 3, 1, 1, act="sparse"),
    LayerCase("special_vals", 2, 64, 8, 8, 16, 3, 1, 1, act="special"),
    # channel tails / ragged sizes / non-multiple-of-32 outputs
    LayerCase("tail_c3", 2, 3, 11, 13, 16, 3, 1, 1, act="normal", winit="default", bias=True),
    LayerCase("tail_c16_1x1", 2, 16, 8, 8, 16, 1, 1, 0, act="relu", winit="default", bias=True, post="scale"),
    LayerCase("tail_c96", 1, 96, 10, 10, 40, 3, 1, 1, act="relu"),
    LayerCase("tail_c200_o5", 1, 200, 7, 5, 5, 3, 2, 1, act="normal"),
    LayerCase("pad0_3x3", 2, 64, 10, 10, 32, 3, 1, 0, act="relu"),
    LayerCase("k5_generic", 1, 32, 9, 9, 8, 5, 1, 2, act="normal"),
    LayerCase("k3_dil2_generic", 1, 64, 12, 12, 16, 3, 1, 2, act="relu", dilation=2),
    LayerCase("k1_s2", 2, 128, 9, 9, 64, 1, 2, 0, act="relu"),
    LayerCase("c1024_1x1", 1, 1024, 4, 4, 64, 1, 1, 0, act="relu"),
    # weight binarizer options
    LayerCase("center", 2, 64, 8, 8, 32, 3, 1, 1, act="relu", center=True),
    LayerCase("no_alpha", 2, 64, 8, 8, 32, 3, 1, 1, act="relu", compute_alpha=False, post="scale"),
    LayerCase("center_bias_scale", 2, 128, 6, 6, 48, 3, 1, 1, act="normal", center=True, bias=True, post="scale"),
    LayerCase("zero_weights", 2, 64, 8, 8, 32, 3, 1, 1, act="relu", winit="withzeros"),
]

LAYER_CASES_BY_NAME = {c.name: c for c in LAYER_CASES}


@dataclass(frozen=True)
class LinearCase:
    name: str
    B: int
    F: int
    O: int
    act: str = "normal"
    bias: bool = True
    post: str = "identity"
    center: bool = False

    def tensors(self):
        s = gen.seed_of("linear", self.name)
        x = gen.activation(self.act, s, (self.B, self.F, 1, 1))[:, :, 0, 0]
        w = gen.conv_weight("default", s + 11, (self.O, self.F))
        b = (0.1 * gen.normal(s + 12, (self.O,))).astype(np.float32) if self.bias else None
        sc = (0.5 + gen.uniform(s + 13, (self.O,))).astype(np.float32) if self.post == "scale" else None
        return x, w, b, sc


LINEAR_CASES = [
    LinearCase("fc_512_1000", 4, 512, 1000, act="relu"),
    LinearCase("fc_10_3", 3, 10, 3, act="normal", post="scale"),
    LinearCase("fc_100_70_center", 5, 100, 70, act="sparse", center=True),
]


def grad_case(name: str) -> LayerCase:
    """The layer cases of the gradient fixtures (tests/golden/grads.npz, generated from the reference's autograd):
    N = 2, a per-channel post scale, at most 14 x 14 pixels."""
    import dataclasses
    c = LAYER_CASES_BY_NAME[name]
    return dataclasses.replace(c, N=2, post="scale", H=min(c.H, 14), W=min(c.W, 14))
