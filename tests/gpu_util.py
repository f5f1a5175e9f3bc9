"""Helpers shared by the -m gpu parity tests (they call the product through its C-ABI / module API and
compare with the CPU oracle)."""
import numpy as np
import torch


def nhwc(x, dtype):
    """NCHW cpu fp32 -> NHWC cuda tensor of the activation dtype."""
    return x.permute(0, 2, 3, 1).contiguous().to("cuda").to(dtype)


def hp_pack(x, scale=1.0):
    """fp32 cuda tensor whose innermost extent is a multiple of 64 -> the half-pair storage of PH_PREC_FP16X3 (same shape and
    byte size, an opaque float32 container; include/pathomic_hip.h: ph_hp_pack)."""
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib().ph_hp_pack(ptr(x), ptr(out), x.numel(), float(scale), stream()), "ph_hp_pack")
    return out


def hp_unpack(x):
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    out = torch.empty_like(x)
    check(lib().ph_hp_unpack(ptr(x), ptr(out), x.numel(), stream()), "ph_hp_unpack")
    return out


def nchw_cpu(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def _d(a):
    if torch.is_tensor(a):
        return a.detach().double().cpu()
    return torch.as_tensor(np.asarray(a)).double()


def maxerr(a, b):
    a, b = _d(a), _d(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs().max().item(), a.abs().max().item()


def assert_close(ref, got, atol, rtol=0.0, what=""):
    err, mx = maxerr(ref, got)
    assert err <= atol + rtol * mx, f"{what}: max|err| {err:.3e} > {atol:.1e} + {rtol:.1e}*{mx:.3e}"
    return err


class Report:
    """Collects every comparison of a test, prints the whole table, fails at the end if any exceeded its
    tolerance (one GPU round trip shows all errors instead of the first)."""

    def __init__(self, title):
        self.title, self.rows = title, []

    def close(self, ref, got, atol, rtol=0.0, what=""):
        err, mx = maxerr(ref, got)
        self.rows.append((what, err, mx, atol + rtol * mx))
        return err

    def finish(self):
        print(f"\n== {self.title}")
        bad = []
        for what, err, mx, tol in self.rows:
            flag = "" if err <= tol else "   <-- FAIL"
            print(f"   {what:<34s} max|err| {err:9.3e}   max|ref| {mx:9.3e}   tol {tol:9.3e}{flag}")
            if err > tol:
                bad.append(what)
        assert not bad, f"{self.title}: out of tolerance: {bad}"
