import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/binary-networks-pytorch_amd"]
import torch
from bnn_amd import hipops
DEV="cuda:0"
g = torch.Generator().manual_seed(0)
c_in, planes, hw, N = 256, 256, 14, 2
waves = int(sys.argv[1])
ws = [torch.randn(planes // 2, c_in, 3, 3, generator=g).to(DEV), torch.randn(planes // 4, planes // 2, 3, 3, generator=g).to(DEV), torch.randn(planes // 4, planes // 4, 3, 3, generator=g).to(DEV)]
bn = lambda c: ((torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.3).to(DEV))
x = torch.randn(N, c_in, hw, hw, generator=g).to(DEV)
res = torch.randn(N, planes, hw, hw, generator=g).to(DEV)
p_in = hipops.bn_act_pack(x, *bn(c_in), relu=True)
pack = hipops.hblock_pack(*[hipops.pack_weight(w) for w in ws], bn(planes // 2), bn(planes // 4), bn(planes))
want_y, want_p = hipops.hblock_forward(p_in, pack, res)
torch.cuda.synchronize()
print("ref done", flush=True)
y, p = hipops.hblock_forward(p_in, pack, res, channel_lanes=True, waves=waves)
torch.cuda.synchronize()
print("cl done waves", waves, "y equal", torch.equal(y, want_y), "p equal", torch.equal(p.P, want_p.P), float((y-want_y).abs().max()), flush=True)
pack_last = hipops.hblock_pack(*[hipops.pack_weight(w) for w in ws], bn(planes // 2), bn(planes // 4), None)
# (bn() draws new constants: compare against the pixel-lane kernel with the SAME pack)
want2, _ = hipops.hblock_forward(p_in, pack_last, res, out_packed=False)
for trial in range(3):
    y2, _ = hipops.hblock_forward(p_in, pack_last, res, out_packed=False, channel_lanes=True, waves=waves)
    torch.cuda.synchronize()
    bad = (y2 != want2)
    print("no-next trial", trial, "mismatches", int(bad.sum()), "per channel-quarter", [int(bad[:, a:b].sum()) for a, b in ((0, planes // 2), (planes // 2, 3 * planes // 4), (3 * planes // 4, planes))],
          "rows with mismatches", sorted(set(bad.nonzero()[:, 2].tolist()))[:20], "images", sorted(set(bad.nonzero()[:, 0].tolist())))
