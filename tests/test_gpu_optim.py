"""The fused Adam + EMA kernel (ph_adam_ema_step_dev; reference networks_new.py:85 torch.optim.Adam and
train_test_path_multi_distill.py:34-38 update_ema_variables) in isolation against torch.optim.Adam on the CPU, fed the
SAME gradients: the update itself, decoupled from whatever produced the gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_adam_and_ema_equal_torch_adam_given_the_same_gradients():
    import multimodal_learning_amd as m
    torch.manual_seed(3)
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (3, 128), (3,), (5,)]      # sizes that are not multiples of 4 included
    ref_p = [torch.nn.Parameter(torch.randn(s) * 0.1) for s in shapes]
    ref_ema = [p.detach().clone() * 0.9 + 0.01 for p in ref_p]
    dev_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ema_p = [torch.nn.Parameter(e.clone().cuda(), requires_grad=False) for e in ref_ema]
    kw = dict(lr=5e-4, betas=(0.9, 0.999), weight_decay=4e-4)
    ref_opt = torch.optim.Adam(ref_p, **kw)
    opt = m.FusedAdam(dev_p, **kw)
    flat = opt.flat
    opt.ema_flat = m.FlatParams(ema_p)
    opt.ema_range = (0, flat.numel)
    # a mid-training optimiser state in torch's own layout, imported by both
    state = {i: dict(step=torch.tensor(11.0), exp_avg=torch.randn(s) * 1e-3, exp_avg_sq=torch.rand(s) * 1e-5 + 1e-9)
             for i, s in enumerate(shapes)}
    sd = ref_opt.state_dict()
    sd["state"] = {i: {k: v.clone() for k, v in st.items()} for i, st in state.items()}
    ref_opt.load_state_dict(sd)
    sd2 = opt.state_dict(); sd2.pop("fused"); sd2["state"] = state
    opt.load_state_dict(sd2)
    for it in range(3):
        grads = [torch.randn(s) * (10.0 ** -(it + i % 3)) for i, s in enumerate(shapes)]
        grads[1][:] = 0.0                                    # an all-zero gradient: weight decay and momentum only
        for p, q, g in zip(ref_p, dev_p, grads):
            p.grad = g.clone()
        opt.zero_grad()
        for q, g in zip(dev_p, grads):
            q.grad.copy_(g)
        alpha = min(1 - 1 / (it + 12), 0.99)
        opt.ema_alpha = alpha
        ref_opt.step()
        opt.step()
        for e, p in zip(ref_ema, ref_p):
            e.mul_(alpha).add_(p.detach(), alpha=1 - alpha)
        for i, (p, q, e, f) in enumerate(zip(ref_p, dev_p, ref_ema, ema_p)):
            dp = (p.detach() - q.detach().cpu()).abs().max().item()
            de = (e - f.detach().cpu()).abs().max().item()
            assert dp <= 2e-8 + 2e-7 * p.detach().abs().max().item(), (it, i, dp)
            assert de <= 2e-8 + 2e-7 * e.abs().max().item(), (it, i, de)
    # exported state (torch layout) equals torch's
    out = opt.state_dict()["state"]
    for i, p in enumerate(ref_p):
        st = ref_opt.state[p]
        assert int(out[i]["step"]) == int(st["step"]) == 14
        for key in ("exp_avg", "exp_avg_sq"):       # (fused multiply-add vs two roundings: a few ulp of the larger term)
            a, b = out[i][key].cpu().numpy(), st[key].numpy()
            assert np.abs(a - b).max() <= 4e-7 * np.abs(b).max() + 1e-20, (i, key, np.abs(a - b).max(), np.abs(b).max())
    # and a reload of the same checkpoint object works twice (load_state_dict must not consume its argument)
    ck = opt.state_dict()
    opt.load_state_dict(ck); opt.load_state_dict(ck)
    assert "fused" in ck and opt._step == 14


def test_fused_adagrad_and_ema_equal_torch_adagrad_given_the_same_gradients():
    """VERDICT r04 missing 3: define_optimizer's `adagrad` branch (networks_new.py:86-87: torch.optim.Adagrad(lr, weight_decay,
    initial_accumulator_value=0.1)) as the fused kernel ph_adagrad_ema_step_dev, against torch.optim.Adagrad on the CPU fed the
    same gradients for three steps; the EMA copy against update_ema_variables' rule; state_dict in torch's layout."""
    import multimodal_learning_amd as m
    from types import SimpleNamespace
    torch.manual_seed(4)
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (3, 128), (3,), (5,)]
    ref_p = [torch.nn.Parameter(torch.randn(s) * 0.1) for s in shapes]
    ref_ema = [p.detach().clone() * 0.9 + 0.01 for p in ref_p]
    holder = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p])
    ema_p = [torch.nn.Parameter(e.clone().cuda(), requires_grad=False) for e in ref_ema]
    opt_ns = SimpleNamespace(optimizer_type="adagrad", lr=2e-3, weight_decay=4e-4, beta1=0.9, beta2=0.999)
    opt = m.networks_new.define_optimizer(opt_ns, holder)
    assert type(opt).__name__ == "FusedAdagrad"
    ref_opt = torch.optim.Adagrad(ref_p, lr=2e-3, weight_decay=4e-4, initial_accumulator_value=0.1)
    dev_p = list(holder)
    flat = opt.flat
    opt.ema_flat = m.FlatParams(ema_p)
    opt.ema_range = (0, flat.numel)
    for it in range(3):
        grads = [torch.randn(s) * (10.0 ** -(it + i % 3)) for i, s in enumerate(shapes)]
        grads[1][:] = 0.0
        for p, g in zip(ref_p, grads):
            p.grad = g.clone()
        opt.zero_grad()
        for q, g in zip(dev_p, grads):
            q.grad.copy_(g)
        alpha = min(1 - 1 / (it + 12), 0.99)
        opt.ema_alpha = alpha
        ref_opt.step()
        opt.step()
        for e, p in zip(ref_ema, ref_p):
            e.mul_(alpha).add_(p.detach(), alpha=1 - alpha)
        for i, (p, q, e, f) in enumerate(zip(ref_p, dev_p, ref_ema, ema_p)):
            dp = (p.detach() - q.detach().cpu()).abs().max().item()
            de = (e - f.detach().cpu()).abs().max().item()
            assert dp <= 2e-8 + 2e-7 * p.detach().abs().max().item(), (it, i, dp)
            assert de <= 2e-8 + 2e-7 * e.abs().max().item(), (it, i, de)
    out = opt.state_dict()["state"]
    for i, p in enumerate(ref_p):
        st = ref_opt.state[p]
        assert int(out[i]["step"]) == int(st["step"]) == 3
        a, b = out[i]["sum"].cpu().numpy(), st["sum"].numpy()
        assert np.abs(a - b).max() <= 4e-7 * np.abs(b).max(), (i, np.abs(a - b).max())
    # round trip through torch's own layout
    sd = ref_opt.state_dict()
    opt2 = m.networks_new.define_optimizer(opt_ns, torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in dev_p]))
    opt2.load_state_dict(sd)
    back = opt2.state_dict()["state"]
    assert opt2._step == 3 and all(torch.equal(back[i]["sum"].cpu(), ref_opt.state[p]["sum"]) for i, p in enumerate(ref_p))
    with pytest.raises(NotImplementedError):
        m.networks_new.define_optimizer(SimpleNamespace(optimizer_type="adabound", lr=1e-3, weight_decay=0.0, beta1=0.9, beta2=0.999), holder)


def test_fused_adam_follows_a_cycled_beta1_inside_a_captured_graph():
    """`--lr_policy onecycle` (networks_new.py:124-125) cycles beta1 with the learning rate: FusedAdam hands lr AND the betas
    to its kernel through device memory, so a step replayed from a captured HIP graph follows the schedule - against
    torch.optim.Adam driven by the same OneCycleLR on the same gradients."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt
    torch.manual_seed(0)
    opt = default_opt(lr_policy="onecycle", niter=1, niter_decay=1)
    ref_p = [torch.randn(300, 7, device="cuda").requires_grad_(True), torch.randn(41, device="cuda").requires_grad_(True)]
    our_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=opt.lr, betas=(opt.beta1, opt.beta2), weight_decay=opt.weight_decay)
    ours = m.train_step.FusedAdam(our_p, lr=opt.lr, betas=(opt.beta1, opt.beta2), weight_decay=opt.weight_decay)
    sr = torch.optim.lr_scheduler.OneCycleLR(ref, max_lr=1e-3, epochs=2, steps_per_epoch=200)
    so = m.networks_new.define_scheduler(opt, ours)
    ours.zero_grad()
    gs = [torch.randn(60, *p.shape, device="cuda") for p in ref_p]
    gbuf = [torch.zeros_like(p) for p in our_p]
    graph = None
    for it in range(60):
        for p, pr, g_, gb in zip(our_p, ref_p, gs, gbuf):
            pr.grad = g_[it].clone()
            gb.copy_(g_[it])
        ours.prepare_step()
        if it < 3:
            for p, gb in zip(our_p, gbuf):
                p.grad.copy_(gb)
            ours.step()
        else:
            if graph is None:
                graph = torch.cuda.CUDAGraph()
                ours._prepared = True
                with torch.cuda.graph(graph):
                    for p, gb in zip(our_p, gbuf):
                        p.grad.copy_(gb)
                    ours.step()
            ours._prepared = False
            graph.replay()
        ref.step()
        sr.step(); so.step()
        assert ours.param_groups[0]["betas"][0] == ref.param_groups[0]["betas"][0]
    assert ref.param_groups[0]["betas"][0] != opt.beta1          # (the schedule did move beta1)
    for p, pr in zip(our_p, ref_p):
        assert (p - pr).abs().max().item() <= 2e-6 * pr.abs().max().item()
