"""CPU restatement (test infrastructure only) of the reference's contrast-index sampling rule, numpy RNG as in the
reference: MICCAI-2022/data_loaders_MT.py:187-203 (class lists) and :229-249 (per-item draw); MIA-2023 `neg_mode`
("MIA 2023/stage2_unimodal_student/data_loaders_MT.py":218-238).  PINNED: tests/golden/make_golden_sampler.py imports the
reference's dataset class (stub `torchvision`: its transforms never touch numpy's stream), calls `__getitem__` for a fixed
index sequence under a fixed `np.random.seed`, and tests/test_oracle_golden.py::test_sampler_draws_equal_the_reference holds
this file to those rows draw for draw (11 cases: both trainers, every pos_mode / neg_mode, K below and above the candidate
list).  The GPU sampler (csrc/sampler.hip) uses its own counter-based stream and is compared with this rule distributionally."""
import numpy as np


def class_lists(labels, num_classes):
    labels = np.asarray(labels).astype(int)
    cls_positive = [[] for _ in range(num_classes)]
    for i in range(len(labels)):
        cls_positive[labels[i]].append(i)
    cls_negative = [[] for _ in range(num_classes)]
    for i in range(num_classes):
        for j in range(num_classes):
            if j == i:
                continue
            cls_negative[i].extend(cls_positive[j])
    return [np.asarray(c) for c in cls_positive], [np.asarray(c) for c in cls_negative]


def sample_item(rng, index, g, cls_positive, cls_negative, n_data, P, K, pos_mode="multi_pos", neg_mode="diff_class"):
    """One __getitem__ draw (:229-249).  `rng`: a numpy RandomState standing for the worker's global np.random."""
    if pos_mode == "exact":
        pos_idx = np.asarray([index])
    elif pos_mode == "relax":
        pos_idx = np.asarray([rng.choice(cls_positive[g], 1)[0]])
    elif pos_mode == "multi_pos":
        pos_idx = rng.choice(cls_positive[g], P, replace=False)
        pos_idx[0] = index
    else:
        raise NotImplementedError(pos_mode)
    if neg_mode == "all_others":
        all_neg_idx = list(range(0, n_data))
        all_neg_idx.remove(index)
        neg_idx = rng.choice(all_neg_idx, K, replace=K > len(all_neg_idx))
    elif neg_mode == "diff_class":
        neg_idx = rng.choice(cls_negative[g], K, replace=K > len(cls_negative[g]))
    else:
        raise NotImplementedError(neg_mode)
    return np.hstack((pos_idx, neg_idx))
