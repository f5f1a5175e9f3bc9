import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import hp_pack, hp_unpack
g = torch.Generator(device="cuda").manual_seed(1)
for scale in (1.0, 0.03, 500.0, 1e-3):
    x = (torch.rand(64, 128, 128, 64, device="cuda", generator=g) * 2 - 1) * scale
    y = hp_unpack(hp_pack(x))
    d = (y - x).abs()
    bad = d > 2.0 ** -20 * scale
    print("scale %g: n %d  bad %d  max|d| %.3e" % (scale, x.numel(), int(bad.sum()), d.max().item()))
    if bad.any():
        i = bad.flatten().nonzero()[:5].flatten()
        for j in i.tolist():
            print("   idx %d  x %.8e (%s)  y %.8e" % (j, x.flatten()[j].item(), hex(x.flatten()[j].view(torch.int32).item() & 0xffffffff), y.flatten()[j].item()))
