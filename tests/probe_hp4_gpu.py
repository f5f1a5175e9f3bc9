"""Debug aid: mid-state golden, step 0 in bf16x6 and fp16x3: gradient of conv1.weight vs golden, and ReLU-mask differences
between the two arithmetics in the student's forward."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import multimodal_learning_amd as m
from multimodal_learning_amd._lib import lib, check
from oracle import weights as W
from oracle.step import default_opt, synthetic_batch
from tests.test_gpu_step import _mk_step, _tuple
from tests.gpu_util import hp_unpack

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "midstate_b8_h96.npz"))
seed, n_data, t0 = int(g["seed"]), int(g["n_data"]), int(g["t0"])
L = lib()


def run(mode):
    m.set_precision(mode)
    step = _mk_step(default_opt(), n_data, seed=seed)
    names = ["student." + k for k, _ in step.model.named_parameters()]
    for i in range(2):
        names += [f"crd{i}.embed_s.linear.weight", f"crd{i}.embed_s.linear.bias", f"crd{i}.embed_t.linear.weight", f"crd{i}.embed_t.linear.bias"]
    params = list(step.module_list.parameters())
    scales = dict(zip([str(s) for s in g["scale_names"]], g["scale_values"]))
    trainable = [(n, tuple(p.shape)) for n, p in zip(names, params) if p.requires_grad]
    mom = W.adam_moments(trainable, scales, seed + 30)
    sd = step.optimizer.state_dict(); sd.pop("fused")
    sd["state"] = {i: dict(step=torch.tensor(float(t0)), exp_avg=mom[n][0], exp_avg_sq=mom[n][1]) for i, (n, p) in enumerate(zip(names, params)) if p.requires_grad}
    step.optimizer.load_state_dict(sd)
    step.iter_num = t0
    for crd, key in ((step.criterion_kd, "Z0"), (step.criterion_kd_path, "Z1")):
        crd.contrast.params[2:4] = torch.as_tensor(g[key]).cuda(); crd.contrast._z_set = True
    named = dict(step.model.named_parameters())
    grads = {}
    orig = step.optimizer.step

    def spy(*a, **k):
        if not grads:
            for kk, p in named.items():
                if p.grad is not None:
                    grads[kk] = p.grad.detach().clone()
        return orig(*a, **k)
    step.optimizer.step = spy
    bt = synthetic_batch(int(g["B"]), int(g["H"]), n_data=n_data, seed=310)
    step.step(_tuple(bt), epoch=int(g["epoch"]), ranks=[g["ranks"][0], g["ranks"][1]])
    torch.cuda.synchronize()
    net = step.model
    key, ws = list(net._ws_cache.items())[-1]
    plan = net._plans[key]
    acts = {}
    for what, nm, n in ((1, "out", 8), (2, "a1", 8), (3, "pool", 1)):
        for i in range(n):
            off = C.c_size_t(0); dims = (C.c_int * 4)()
            check(L.ph_resnet_tensor_info(plan.h, what, i, C.byref(off), dims), "ti")
            d = tuple(dims); cnt = d[0] * d[1] * d[2] * d[3]
            t = ws[off.value: off.value + 4 * cnt].view(torch.float32).view(*d).clone()
            if mode == "fp16x3":
                t = hp_unpack(t.contiguous())
            acts[(nm, i)] = t
    return grads, acts


ga, aa = run("bf16x6")
gb, ab = run("fp16x3")
ref = torch.as_tensor(g["g0_conv1.weight"]).cuda()
for nm, gg in (("bf16x6", ga), ("fp16x3", gb)):
    d = (gg["conv1.weight"].reshape(-1)[:4096] - ref).abs().max().item()
    print(nm, "grad conv1.weight vs golden: max|err| %.3e (max|ref| %.3e)" % (d, ref.abs().max().item()))
print("x6 vs hp grad conv1.weight: %.3e" % (ga["conv1.weight"] - gb["conv1.weight"]).abs().max().item())
for k in aa:
    ma, mb = aa[k] > 0, ab[k] > 0
    nf = int((ma != mb).sum())
    dv = (aa[k] - ab[k]).abs().max().item()
    if nf:
        idx = (ma != mb).nonzero()
        vals = [(aa[k][tuple(i)].item(), ab[k][tuple(i)].item()) for i in idx[:4]]
        print(k, "mask flips:", nf, " max|d act| %.2e" % dv, " values (x6, hp):", vals)
print("activation tensors compared:", len(aa))
m.set_precision("bf16")
