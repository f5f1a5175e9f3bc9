"""Debug aid (not a pytest file): student forward + backward in bf16x6 and fp16x3 on the same weights / input, per-parameter
gradient differences and the error pattern of conv1.weight."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodal_learning_amd as m
from oracle import weights as W
from oracle.step import default_opt


def run(mode, B, H):
    m.set_precision(mode)
    net = m.define_net(default_opt(), 1, path_only=True)
    net.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    net = net.cuda().train()
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(B, 3, H, H, generator=g) * 2 - 1).cuda()
    f3, feat, hazard, pred, _ = net(x_path=x)
    loss = (feat * torch.linspace(0.5, 1.5, 128).cuda()).sum() + (hazard * torch.tensor([1.0, -2.0, 0.5]).cuda()).sum() + 0.1 * f3.sum()
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}, hazard.detach().clone()


B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 96
ga, ha = run("bf16x6", B, H)
gb, hb = run("fp16x3", B, H)
print("hazard diff", (ha - hb).abs().max().item())
for k in ga:
    d = (ga[k] - gb[k]).abs().max().item()
    mx = ga[k].abs().max().item()
    if d > 1e-4 * mx:
        print("%-40s max|d| %.3e  max|ref| %.3e  rel %.2e" % (k, d, mx, d / mx))
d = (ga["conv1.weight"] - gb["conv1.weight"]).abs()
print("conv1.weight err by kh:", [round(v, 5) for v in d.amax(dim=(0, 1, 3)).tolist()])
print("conv1.weight err by kw:", [round(v, 5) for v in d.amax(dim=(0, 1, 2)).tolist()])
print("conv1.weight err by ch:", [round(v, 5) for v in d.amax(dim=(0, 2, 3)).tolist()])
print("conv1.weight ref by kh:", [round(v, 3) for v in ga["conv1.weight"].abs().amax(dim=(0, 1, 3)).tolist()])
m.set_precision("bf16")
