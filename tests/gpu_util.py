"""Helpers shared by the -m gpu parity tests (they call the product through its C-ABI / module API and
compare with the CPU oracle)."""
import numpy as np
import torch


def nhwc(x, dtype):
    """NCHW cpu fp32 -> NHWC cuda tensor of the activation dtype."""
    return x.permute(0, 2, 3, 1).contiguous().to("cuda").to(dtype)


def nchw_cpu(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def maxerr(a, b):
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs().max().item(), a.abs().max().item()


def assert_close(ref, got, atol, rtol=0.0, what=""):
    err, mx = maxerr(ref, got)
    assert err <= atol + rtol * mx, f"{what}: max|err| {err:.3e} > {atol:.1e} + {rtol:.1e}*{mx:.3e}"
    return err
