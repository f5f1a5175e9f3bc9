"""On-device contrast-index sampler: the `sample_idx` rule of the reference's dataset class
`Pathomic_InstanceSample` (MICCAI-2022/data_loaders_MT.py:146-256; MIA-2023 `neg_mode`,
"MIA 2023/stage2_unimodal_student/data_loaders_MT.py":205-238) as one kernel launch per batch instead of
`np.random.choice` calls inside the DataLoader workers.  The class lists are built exactly as the reference builds
them (:187-203); the draws are distributionally - not bit-for-bit - equal to numpy's (see csrc/sampler.hip)."""
import numpy as np
import torch

from ._lib import lib, check, ptr, stream

_POS = {"exact": 0, "relax": 1, "multi_pos": 2}
_NEG = {"diff_class": 0, "all_others": 1}


class ContrastIndexSampler:
    """sampler = ContrastIndexSampler(opt, labels, device); sample_idx = sampler(index, grade)  -> int64 [B, np + K]
    with np = opt.nce_p for pos_mode 'multi_pos' and 1 otherwise; column 0 is the query's own index for 'exact' and
    'multi_pos' (:231, :239).  Reads opt.nce_p, opt.nce_k, opt.pos_mode and (MIA-2023) opt.neg_mode."""

    def __init__(self, opt, labels, device="cuda", seed=0, num_classes=None):
        labels = np.asarray(labels).astype(np.int64)
        self.n_data = int(labels.shape[0])
        self.num_classes = int(num_classes if num_classes is not None else getattr(opt, "label_dim", labels.max() + 1))
        self.P, self.K = int(opt.nce_p), int(opt.nce_k)
        self.pos_mode = getattr(opt, "pos_mode", "multi_pos")
        self.neg_mode = getattr(opt, "neg_mode", "diff_class")
        if self.pos_mode not in _POS:
            raise NotImplementedError(self.pos_mode)                       # :240-241
        if self.neg_mode not in _NEG:
            raise NotImplementedError("neg_mode '%s' ('both_models' concatenates the two lists)" % self.neg_mode)
        # :187-203 - rows of every class in ascending order; negatives of class i = the other classes' lists in order
        cls_positive = [np.nonzero(labels == c)[0] for c in range(self.num_classes)]
        cls_negative = [np.concatenate([cls_positive[j] for j in range(self.num_classes) if j != i]) if self.num_classes > 1
                        else np.zeros(0, dtype=np.int64) for i in range(self.num_classes)]
        if self.pos_mode == "multi_pos" and min(len(c) for c in cls_positive) < self.P:
            raise ValueError("multi_pos draws nce_p rows without replacement: every class needs >= nce_p rows "
                             "(np.random.choice raises the same, data_loaders_MT.py:238)")
        self.cls_positive, self.cls_negative = cls_positive, cls_negative
        dev = torch.device(device)

        def flat(lists):
            off = np.zeros(len(lists) + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(x) for x in lists])
            cat = np.concatenate(lists).astype(np.int32) if off[-1] else np.zeros(1, dtype=np.int32)
            return torch.as_tensor(cat, device=dev), torch.as_tensor(off, device=dev)
        self._pos, self._pos_off = flat(cls_positive)
        self._neg, self._neg_off = flat(cls_negative)
        self.seed = int(seed)
        self.step = torch.zeros(1, device=dev, dtype=torch.int64)          # device counter: graph-replayable draws
        self.device = dev

    @property
    def width(self):
        return (self.P if self.pos_mode == "multi_pos" else 1) + self.K

    def __call__(self, index, grade, out=None, advance=True):
        index = index.to(self.device).long().contiguous()
        grade = grade.to(self.device).long().contiguous()
        B = index.shape[0]
        if out is None:
            out = torch.empty(B, self.width, device=self.device, dtype=torch.int64)
        check(lib().ph_contrast_sampler(ptr(index), ptr(grade), ptr(self._pos), ptr(self._pos_off), ptr(self._neg),
                                        ptr(self._neg_off), self.n_data, B, self.P, self.K, _POS[self.pos_mode],
                                        _NEG[self.neg_mode], self.seed, ptr(self.step), ptr(out), stream()),
              "ph_contrast_sampler")
        if advance:
            self.step += 1
        return out
