"""Microbenchmark of the t-SVD proximal update (ph_tsvd_update_aux) on the adjacency stacks the stage-1 step really feeds it
(captured from a few steps of bench.py's `tsvd` variant): time per call, Jacobi sweeps per frequency slice, and the error
against a float64 torch.linalg.svd restatement of train_test_tSVD.py:382's update.  Not a pytest file; run on the GPU box:
    python tests/bench_tsvd_gpu.py [B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_learning_amd as m  # noqa: E402
from multimodal_learning_amd import tsvd as T  # noqa: E402
from multimodal_learning_amd._lib import lib, ptr, stream, check  # noqa: E402


def svd_reference(stack, tau):
    """float64: FFT along views, singular-value soft threshold per slice, inverse FFT; TNN = sum of kept values / V."""
    x = torch.fft.fft(stack.double(), dim=2)
    V = x.shape[2]
    y = torch.zeros_like(x)
    tnn = 0.0
    for k in range(V):
        u, s, vh = torch.linalg.svd(x[:, :, k])
        s = torch.clamp(s - tau, min=0)
        tnn += float(s.sum())
        y[:, :, k] = (u * s.to(u.dtype)) @ vh
    return torch.fft.ifft(y, dim=2).real, tnn / V


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    step, bts, _ = bench.variant_setup("tsvd", B, 512, dev)
    caught = []
    orig = T.update_aux

    def spy(stack, tau, print_bool=False):
        caught.append((stack.clone(), float(tau)))
        return orig(stack, tau, print_bool)
    T.update_aux = spy
    for i in range(4):
        step.step(bts[i % 2], epoch=5)
    torch.cuda.synchronize()
    T.update_aux = orig
    L = lib()
    for stack, tau in caught[-2:]:
        Bq, _, V = stack.shape
        a = stack.float().permute(2, 0, 1).contiguous()
        aux = torch.empty_like(a)
        tnn = torch.empty(1, device=dev)
        nws = L.ph_tsvd_workspace_bytes(V, Bq)
        ws = torch.zeros(nws, device=dev, dtype=torch.uint8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            check(L.ph_tsvd_update_aux(ptr(a), ptr(aux), ptr(tnn), V, Bq, tau, ptr(ws), stream()), "ph_tsvd_update_aux")
        e0.record()
        for _ in range(5):
            check(L.ph_tsvd_update_aux(ptr(a), ptr(aux), ptr(tnn), V, Bq, tau, ptr(ws), stream()), "ph_tsvd_update_aux")
        e1.record()
        torch.cuda.synchronize()
        tk = ws.view(torch.float32)[-16:].cpu()
        ref, rtnn = svd_reference(stack.cpu(), tau)
        got = aux.permute(1, 2, 0).cpu().double()
        err = float((got - ref).abs().max())
        sv = torch.linalg.svdvals(stack[:, :, 0].double().cpu())
        print("B %d V %d tau %.3g  %.1f us/call  sweeps %s  max|aux-ref| %.3g (max|ref| %.3g)  TNN %.6g ref %.6g  "
              "view0 sigma max %.3g min %.3g" % (Bq, V, tau, e0.elapsed_time(e1) * 200, tk[8:8 + V // 2 + 1].tolist(), err,
                                                 float(ref.abs().max()), float(tnn), rtnn, float(sv[0]), float(sv[-1])))


if __name__ == "__main__":
    main()
