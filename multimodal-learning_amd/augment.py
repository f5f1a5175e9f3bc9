"""On-device input pipeline: the training transform of the reference's loader
(MICCAI-2022/data_loaders_MT.py:168-175 run twice by `TransformTwice`, :51-53) on uint8 source tiles that stay resident in
HBM - at the step rates of this package the PIL pipeline in four DataLoader workers is the bottleneck.

    aug = DeviceAugment(opt, device)                    # opt.input_size_path = crop size
    x_path, ema_x_path = aug(src_u8)                    # src_u8 [B, SH, SW, 3] uint8 (device) -> two f32 [B, 3, S, S] views

The draws come from a counter RNG keyed by (seed, step, image, view) - reproducible, not numpy's stream.  The colour
arithmetic is Pillow's as torchvision drives it, pinned bit for bit against Pillow (tests/golden/colorjitter_pil.npz)."""
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

    def __call__(self, src_u8, params=None, rows=None, out=None):
        """`rows` (int64 [B], device): take the batch straight out of a resident tile store src_u8 [n, SH, SW, 3] - no
        gather copy of the 3 MB source tiles."""
        src = require_cuda(src_u8, "src_u8")
        if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3:
            raise RuntimeError("src_u8 must be uint8 [B, SH, SW, 3]")
        src = src.contiguous()
        _, SH, SW, _ = src.shape
        if rows is not None:
            rows = rows.to(src.device).long().contiguous()
        B = src.shape[0] if rows is None else rows.shape[0]
        S = self.S
        if params is None:
            params = torch.empty(B, 2, NPARAM, device=src.device, dtype=torch.float32)
            check(lib().ph_augment_params(ptr(params), B, self.seed, ptr(self.step), SH, SW, S, *self.jitter, stream()),
                  "ph_augment_params")
            self.step += 1
        else:
            # Caller-supplied parameters (tests, replaying a recorded draw; not the hot path, so a host check is fine).
            # Layout [B, 2, NPARAM]: 0-11 flips / crop / jitter factors / step order (oracle/augment.py::draw_params),
            # 12 and 14-15 scratch (zeroed here), 13 = BIT MASK of disabled colour steps (bit k set: ColorJitter dropped
            # step k because its range is zero) - the kernel casts it to int, so it must be an integer in 0..15.
            params = params.to(src.device).float().contiguous().clone()
            c13 = params[..., 13]
            if not bool(((c13 >= 0) & (c13 <= 15) & (c13 == c13.round())).all().item()):
                raise ValueError("augmentation params column 13 is the mask of disabled colour steps: integers 0..15 "
                                 "(got values outside that set - a parameter block from before the layout change?)")
            params[..., 12] = 0
            params[..., 14:] = 0          # the grey-sum accumulator slot (column 13 = mask of disabled steps, kept)
        if out is not None:          # caller-owned buffers (the resident input sets a captured step graph reads from)
            out0, out1 = out
            for t in (out0, out1):
                if tuple(t.shape) != (B, 3, S, S) or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                    raise RuntimeError("out must be two contiguous f32 [B, 3, S, S] device tensors")
        else:
            out0 = torch.empty(B, 3, S, S, device=src.device, dtype=torch.float32)
            out1 = torch.empty_like(out0)
        check(lib().ph_augment_apply(ptr(src), ptr(rows), ptr(params), ptr(out0), ptr(out1), B, SH, SW, S, stream()),
              "ph_augment_apply")
        self.last_params = params
        return out0, out1


class ResidentTileLoader:
    """The training loader's batch tuple (data_loaders_MT.py:256) produced on the device: uint8 tiles, omic vectors and
    labels live in HBM; a batch of row indices becomes ((x_path, ema_x_path), 0, x_omic, 0, 0, grade, index, sample_idx)
    with both augmented views (DeviceAugment, straight from the store) and the contrast indices (ContrastIndexSampler)."""

    def __init__(self, opt, tiles_u8, x_omic, grade, device="cuda", seed=0):
        from .sampler import ContrastIndexSampler
        self.device = torch.device(device)
        self.tiles = require_cuda(tiles_u8, "tiles_u8").contiguous()
        self.x_omic = x_omic.to(self.device).float().contiguous()
        self.grade = grade.to(self.device).long().contiguous()
        self.aug = DeviceAugment(opt, self.device, seed)
        self.sampler = ContrastIndexSampler(opt, self.grade.cpu().numpy(), self.device, seed=seed)
        self.row_to_tile = None       # optional int64 [n_rows]: dataset row -> tile of the store (several rows per tile)
        self.batch_no = None

    def next(self, batch_size=None, into=None):
        """The next batch of an endless shuffled run (DataLoader(shuffle=True, drop_last=True)): its row indices are
        drawn on the device from a device-side batch counter, so the whole call - index draw, both augmented views,
        gathers, contrast indices - is a fixed launch sequence that a HIP graph can capture and replay."""
        from . import ops
        if into is None:
            B = int(batch_size)
            index = torch.empty(B, device=self.device, dtype=torch.int64)
        else:
            index = into[6]
            B = index.shape[0]
        if getattr(self, "batch_no", None) is None:
            self.batch_no = torch.zeros(1, device=self.device, dtype=torch.int64)
        check(lib().ph_shuffle_indices(ptr(index), self.grade.shape[0], B, self.aug.seed, ptr(self.batch_no), stream()),
              "ph_shuffle_indices")
        ops.counter_inc(self.batch_no)
        return self.batch(index, into=into, _index_in_place=into is not None)

    def batch(self, index, into=None, _index_in_place=False):
        """`into`: a previously returned batch tuple whose tensors are refilled in place (fixed addresses: a captured
        step graph that adopted them replays without any staging copy)."""
        index = index.to(self.device).long().contiguous()
        if into is None:
            rows = index if getattr(self, "row_to_tile", None) is None else self.row_to_tile[index]
            x_path, ema_x_path = self.aug(self.tiles, rows=rows)
            grade = self.grade[index]
            z = torch.zeros(index.shape[0], device=self.device)
            return ((x_path, ema_x_path), z, self.x_omic[index], z, z, grade, index, self.sampler(index, grade))
        (x_path, ema_x_path), _, x_omic, _, _, grade, idx_buf, sample_idx = into
        rows = index if getattr(self, "row_to_tile", None) is None else self.row_to_tile[index]
        self.aug(self.tiles, rows=rows, out=(x_path, ema_x_path))
        torch.index_select(self.x_omic, 0, index, out=x_omic)
        torch.index_select(self.grade, 0, index, out=grade)
        if not _index_in_place:
            idx_buf.copy_(index)
        self.sampler(idx_buf, grade, out=sample_idx)
        return into
