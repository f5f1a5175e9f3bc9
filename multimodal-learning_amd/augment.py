"""On-device input pipeline: the training transform of the reference's loader
(MICCAI-2022/data_loaders_MT.py:168-175 run twice by `TransformTwice`, :51-53) on uint8 source tiles that stay resident in
HBM - at the step rates of this package the PIL pipeline in four DataLoader workers is the bottleneck.

    aug = DeviceAugment(opt, device)                    # opt.input_size_path = crop size
    x_path, ema_x_path = aug(src_u8)                    # src_u8 [B, SH, SW, 3] uint8 (device) -> two f32 [B, 3, S, S] views

The draws come from a counter RNG keyed by (seed, step, image, view) - reproducible, not numpy's stream.  The colour
arithmetic restates PIL / torchvision, which are absent here: parity unpinned (csrc/augment.hip, oracle/augment.py)."""
import torch

from ._lib import lib, check, ptr, stream, require_cuda

NPARAM = 16


class DeviceAugment:
    def __init__(self, opt, device="cuda", seed=0, brightness=0.1, contrast=0.1, saturation=0.05, hue=0.01):
        self.S = int(opt.input_size_path)
        self.device = torch.device(device)
        self.seed = int(seed)
        self.jitter = (float(brightness), float(contrast), float(saturation), float(hue))     # :172
        self.step = torch.zeros(1, device=self.device, dtype=torch.int64)                      # device-side draw counter
        self.last_params = None

    def __call__(self, src_u8, params=None):
        src = require_cuda(src_u8, "src_u8")
        if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3:
            raise RuntimeError("src_u8 must be uint8 [B, SH, SW, 3]")
        src = src.contiguous()
        B, SH, SW, _ = src.shape
        S = self.S
        if params is None:
            params = torch.empty(B, 2, NPARAM, device=src.device, dtype=torch.float32)
            check(lib().ph_augment_params(ptr(params), B, self.seed, ptr(self.step), SH, SW, S, *self.jitter, stream()),
                  "ph_augment_params")
            self.step += 1
        else:
            params = params.to(src.device).float().contiguous().clone()
            params[..., 12:] = 0          # the grey-sum accumulator slot
        out0 = torch.empty(B, 3, S, S, device=src.device, dtype=torch.float32)
        out1 = torch.empty_like(out0)
        check(lib().ph_augment_apply(ptr(src), ptr(params), ptr(out0), ptr(out1), B, SH, SW, S, stream()), "ph_augment_apply")
        self.last_params = params
        return out0, out1
