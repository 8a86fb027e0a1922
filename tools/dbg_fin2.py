import sys, os
sys.path[:0] = ['.', 'binary-networks-pytorch_amd']
import torch, numpy as np
from bnn_amd import hipops
from tests.golden import gen
out = sys.argv[1]
res = {}
for i, shape in enumerate([(1, 3, 64, 64), (2, 3, 224, 224)]):
    x = torch.from_numpy(gen.normal(5 + i, shape)).cuda(); w = torch.from_numpy(gen.conv_weight("kaiming", 3, (64, 3, 7, 7))).cuda()
    a = torch.rand(64, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) + 0.5
    b = torch.randn(64, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) * 0.3
    y, pk = hipops.stem7x7(x, w, a, b)
    res["y%d" % i] = y.cpu().numpy(); res["p%d" % i] = pk.P.cpu().numpy()
np.savez(out, **res)
