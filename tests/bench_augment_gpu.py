"""Throughput of the on-device input pipeline (not a test): python tests/bench_augment_gpu.py"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_learning_amd as m

B, SH, S = 64, 1024, 512
src = torch.randint(0, 256, (B, SH, SH, 3), dtype=torch.uint8, device="cuda")
aug = m.augment.DeviceAugment(types.SimpleNamespace(input_size_path=S))
for _ in range(3):
    aug(src)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 20
e0.record()
for _ in range(R):
    aug(src)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / R
byt = B * 2 * (2 * S * S * 3 + 3 * S * S * 4)
print(f"augment: {ms * 1e3:.0f} us per batch of {B} tiles x 2 views ({B / ms * 1e3:.0f} tiles/s, {byt / ms / 1e9:.2f} TB/s algorithmic)")
