"""Throughput of the stage-1 mean-teacher step (row f-1; eager launches, not a test): python tests/bench_stage1_gpu.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_learning_amd as m
from bench import make_batch

B, H = 64, 512
m.set_precision("bf16")
opt = m.stage2_opt(dropout_rate=0.1, batch_size=B, cut_fuse_grad=True, num_teachers=2)
opt.pred_distill, opt.KD_weight, opt.CRD_distill, opt.SP_distill, opt.orth_loss, opt.tSVD_loss = 1, 1.0, 0, 0, "False", "False"
st = m.TeacherStage1Step(opt, device="cuda")
bt = make_batch(B, H, 1024, opt, "cuda", 0)
for _ in range(3):
    st.step(bt)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 10
for _ in range(R):
    st.step(bt)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / R * 1e3
print(f"stage-1 step (student PathomicNet fwd+bwd, EMA PathomicNet fwd, 3-branch NLL + pred-KD, Adam+EMA): {ms:.2f} ms at B={B}, {H}x{H} = {B / ms * 1e3:.0f} tiles/s (eager launches)")
