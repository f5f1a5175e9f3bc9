"""Timing of ph_crd_bank_topk (MIA-2023 full-bank KNN) over bank sizes and batch sizes; run under
rocprofv3 --kernel-trace --stats to see the stages (not a pytest file):
    python tests/bench_topk_gpu.py [n_data ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_learning_amd._lib import lib, ptr, stream, check


def run(n, B, NP=6, reps=20):
    L = lib()
    g = torch.Generator().manual_seed(1)
    mem1 = (torch.rand(n, 128, generator=g) - 0.5).cuda(); mem2 = (torch.rand(n, 128, generator=g) - 0.5).cuda()
    labels = torch.randint(0, 3, (n,), generator=g).int().cuda()
    idx = torch.randint(0, n, (B, 5), generator=g).cuda()
    bl = labels[idx[:, 0]].long()
    nb1 = torch.empty(B, NP, dtype=torch.int64, device="cuda"); nb2 = torch.empty_like(nb1)
    s1 = torch.empty(B, NP, device="cuda"); s2 = torch.empty_like(s1)
    ws = torch.empty(L.ph_crd_bank_topk_workspace_bytes(B, n), dtype=torch.uint8, device="cuda")
    call = lambda: check(L.ph_crd_bank_topk(ptr(mem1), ptr(mem2), ptr(labels), ptr(idx), 5, ptr(bl), B, n, NP, 128,
                                            ptr(nb1), ptr(nb2), ptr(s1), ptr(s2), ptr(ws), stream()), "topk")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    print("n_data %7d  B %3d  %.1f us per call (back-to-back launches, includes launch gaps)" % (n, B, e0.elapsed_time(e1) / reps * 1e3))


if __name__ == "__main__":
    ns = [int(a) for a in sys.argv[1:]] or [2048, 16384, 65536, 262144]
    for n in ns:
        for B in (8, 64):
            run(n, B)
