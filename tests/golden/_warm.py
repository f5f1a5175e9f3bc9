"""Shared by the step-golden generators and the GPU tests: a "mid-training" Adam state (tests/golden/make_golden_midstate.py
explains why).  From zero moments Adam's first updates are lr * sign(g) and amplify fp32 rounding into 1e-2 logit
differences, so nothing after step 0 can be compared at 1e-3; with the optimiser started at step count T0 with moments of
the size of the real gradients (recipe: oracle.weights.adam_moments, per-tensor scales measured on the reference's own
first backward and stored in the fixture) every step of a fixture is comparable at the north-star tolerance."""
import torch

T0 = 7
SEED = 900


def param_names(student, n_crd=2):
    names = ["student." + k for k, _ in student.named_parameters()]
    for i in range(n_crd):
        names += [f"crd{i}.embed_s.linear.weight", f"crd{i}.embed_s.linear.bias", f"crd{i}.embed_t.linear.weight",
                  f"crd{i}.embed_t.linear.bias"]
    return names


def grad_scales(names, params, weight_decay):
    """rms of (g + wd * p) per tensor after a backward pass (what Adam's moments average)."""
    out = {}
    for n, p in zip(names, params):
        if p.grad is not None:
            out[n] = max(float((p.grad.double() + weight_decay * p.detach().double()).pow(2).mean().sqrt()), 1e-7)
    return out


def moments(names, params, scales):
    from oracle import weights as W
    tr = [(n, tuple(p.shape)) for n, p in zip(names, params) if p.requires_grad and n in scales]
    return W.adam_moments(tr, scales, SEED)


def set_torch_adam(optimizer, names, params, scales):
    """Reference side: write the state into torch.optim.Adam."""
    mom = moments(names, params, scales)
    for n, p in zip(names, params):
        if n in mom:
            m, v = mom[n]
            optimizer.state[p] = dict(step=torch.tensor(float(T0)), exp_avg=m.to(p.dtype).clone(), exp_avg_sq=v.to(p.dtype).clone())


def load_fused_adam(optimizer, names, params, scales):
    """Product side: the same state through FusedAdam.load_state_dict in torch.optim.Adam's layout."""
    mom = moments(names, params, scales)
    sd = optimizer.state_dict()
    sd.pop("fused", None)
    sd["state"] = {i: dict(step=torch.tensor(float(T0)), exp_avg=mom[n][0], exp_avg_sq=mom[n][1])
                   for i, (n, p) in enumerate(zip(names, params)) if n in mom}
    optimizer.load_state_dict(sd)


def pack_scales(scales):
    import numpy as np
    return dict(scale_names=np.array(list(scales.keys())), scale_values=np.array(list(scales.values()), dtype=np.float64))


def unpack_scales(g):
    return dict(zip([str(s) for s in g["scale_names"]], g["scale_values"]))
