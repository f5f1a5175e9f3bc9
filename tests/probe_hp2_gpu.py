"""Debug aid (not a pytest file): stage-by-stage comparison of the trunk backward between bf16x6 and fp16x3."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodal_learning_amd as m
from multimodal_learning_amd import ops
from multimodal_learning_amd._lib import lib, check, ptr, stream
from oracle import weights as W
from oracle.step import default_opt

B, H = 8, 96
L = lib()


def run(mode, stops):
    m.set_precision(mode)
    net = m.define_net(default_opt(), 1, path_only=True)
    net.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    net = net.cuda().train()
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(B, 3, H, H, generator=g) * 2 - 1).cuda()
    f3, feat, hazard, pred, _ = net(x_path=x)
    ctx = f3.grad_fn
    plan, ws, packed, table = ctx.plan, ctx.ws, ctx.packed, ctx.table
    gen = torch.Generator(device="cuda").manual_seed(3)
    g4 = torch.randn(B, 512, device="cuda", generator=gen)
    g3 = 0.5 * torch.randn(B, 256, device="cuda", generator=gen)
    params = net._trunk_params()
    grads = [torch.zeros_like(p) for p in params]
    ga = ops.void_array([gg.data_ptr() for gg in grads])

    def info(what, idx):
        off = C.c_size_t(0)
        dims = (C.c_int * 4)()
        check(L.ph_resnet_tensor_info(plan.h, what, idx, C.byref(off), dims), "tensor_info")
        return off.value, tuple(dims)
    out = {}
    for stop, what, dims in stops:
        check(L.ph_resnet_backward_debug(plan.h, table, ptr(packed), ptr(ws), ptr(g3), ptr(g4), ga, stop, stream()), "dbg")
        torch.cuda.synchronize()
        off, _ = info(what, 0)
        n = dims[0] * dims[1] * dims[2] * dims[3]
        t = ws[off: off + 4 * n].view(torch.float32).view(*dims).clone()
        out[(stop, what)] = t
    return out


d1 = (B, 24, 24, 64)
# block 1 (layer1.1): stages 48 bn2, 49 wgrad2, 50 dgrad2, 51 bn1, 52 wgrad1, 53 dgrad1
stops = [(47, 4, d1), (47, 5, d1), (50, 7, d1), (51, 6, d1), (53, 4, d1), (53, 5, d1)]
a = run("bf16x6", stops)
b = run("fp16x3", stops)
for k in a:
    ta, tb = a[k], b[k]
    d = (ta - tb).abs()
    print(k, "max|ref| %.3e  max|d| %.3e  mean|d| %.3e  finite %s" % (ta.abs().max().item(), d.max().item(), d.mean().item(), bool(torch.isfinite(tb).all())))
    if d.max() > 1e-3 * ta.abs().max():
        idx = (d > 1e-3 * ta.abs().max()).nonzero()
        print("   bad elements:", idx.shape[0], "of", d.numel(), " rows:", sorted(set(idx[:, 1].tolist()))[:30], " cols:", sorted(set(idx[:, 2].tolist()))[:30],
              " chans:", len(set(idx[:, 3].tolist())))
from tests.gpu_util import hp_unpack
dzb = hp_unpack(b[(51, 6)].contiguous())
dza = a[(51, 6)]
ratio = (dzb.abs().max() / dza.abs().max()).item()
print("dz1: max|x6| %.4e  max|hp stored| %.4e  ratio %.4e (2^%.2f)" % (dza.abs().max().item(), dzb.abs().max().item(), ratio, torch.log2(torch.tensor(ratio)).item()))
sc = 2.0 ** round(torch.log2((dzb.abs().mean() / dza.abs().mean())).item())
d = (dzb / sc - dza).abs()
print("scale 2^%d  max|d| %.3e  at %s  x6 value %.4e hp value %.4e" % (round(torch.log2(torch.tensor(sc)).item()), d.max().item(), (d == d.max()).nonzero()[0].tolist(),
      dza.flatten()[d.argmax()].item(), (dzb / sc).flatten()[d.argmax()].item()))
print("finite stored:", bool(torch.isfinite(dzb).all()), " count > 60000:", int((dzb.abs() > 60000).sum()))
m.set_precision("bf16")
