"""Probe (not a test): device and host time of the pieces of one ResidentTileLoader batch."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_learning_amd as m
dev = "cuda"; B, S, n_data, nt = 64, 512, 1024, 256
tiles = torch.randint(0, 256, (nt, 2 * S, 2 * S, 3), dtype=torch.uint8).to(dev)
opt = types.SimpleNamespace(input_size_path=S, nce_p=300, nce_k=700, pos_mode="multi_pos", label_dim=3)
ld = m.augment.ResidentTileLoader(opt, tiles, torch.randn(n_data, 320), torch.arange(n_data) % 3, device=dev)
tile_of = torch.arange(n_data, device=dev) % nt
bt = ld.batch(torch.arange(B))
def timed(name, fn, R=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(R): fn()
    e1.record(); th = (time.perf_counter() - t0) / R * 1e3; torch.cuda.synchronize()
    print(f"{name:14s} device {e0.elapsed_time(e1) / R:7.3f} ms   host (launch) {th:7.3f} ms")
idx = torch.randperm(n_data, device=dev)[:B]
timed("randperm", lambda: torch.randperm(n_data, device=dev)[:B])
timed("tile_of[idx]", lambda: tile_of[idx])
rows = tile_of[idx]
timed("augment", lambda: ld.aug(ld.tiles, rows=rows, out=bt[0]))
timed("index_select", lambda: (torch.index_select(ld.x_omic, 0, idx, out=bt[2]), torch.index_select(ld.grade, 0, idx, out=bt[5])))
timed("sampler", lambda: ld.sampler(bt[6], bt[5], out=bt[7]))
