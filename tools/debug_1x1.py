import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np, torch, oracle
from bnn_amd import hipops
from tests.golden import gen
for (C, O, k, hw) in [(64, 128, 1, 8), (64, 128, 3, 8), (128, 256, 1, 4), (64, 32, 1, 8)]:
    x = gen.activation('relu', 1, (2, C, hw, hw)); w = gen.conv_weight('kaiming', 2, (O, C, k, k))
    for center in (False, True):
        pw = hipops.pack_weight(torch.from_numpy(w).cuda(), center, True)
        wb, wz, al, anyz = oracle.pack_weight(w, center, True)
        eqb = np.array_equal(pw.wbits.cpu().numpy().view(np.uint32), wb)
        eqa = np.array_equal(pw.alpha.cpu().numpy(), al)
        act = hipops.pack_act(torch.from_numpy(x).cuda())
        out = hipops.bconv2d(act, pw, padding=k // 2).cpu().numpy()
        ref, _ = oracle.binary_conv2d_int(x, w, padding=k // 2, center=center)
        print(C, O, k, "center", center, "bits_eq", eqb, "alpha_eq", eqa, "has_zero", pw.has_zero, anyz, "out_diff", np.abs(out - ref).max())
