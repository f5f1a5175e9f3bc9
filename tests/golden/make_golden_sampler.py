#!/usr/bin/env python3
"""Golden vectors for the contrast-index draw of the reference's dataset class `Pathomic_InstanceSample.__getitem__`
(MICCAI-2022/data_loaders_MT.py:205-256; the `neg_mode` variants of "MIA 2023/stage2_unimodal_student/data_loaders_MT.py":
184-245), produced by importing and RUNNING the reference classes on CPU (build container only).

VERDICT r04 weak 2 / next 9: oracle/sampler.py used to be "pinned by reading" because the dataset module imports torchvision,
which is absent here.  The draw itself only uses numpy's global RNG (`np.random.choice`), so the module is imported with a stub
`torchvision` whose transforms are identities (they never touch numpy's stream - the real ones draw from torch's generator),
fed temporary PNG files, and `__getitem__` is called for a fixed index sequence under a fixed `np.random.seed`.  Equal class
sizes keep `np.asarray(list_of_arrays)` (:201-202) rectangular under numpy 2.  Saved: labels, index sequence, the drawn
`sample_idx` rows per (trainer, pos_mode, neg_mode, P, K) case.

Usage:  python tests/golden/make_golden_sampler.py        # writes tests/golden/sampler_draws.npz
"""
import contextlib
import importlib.util
import io
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG   # noqa: E402

REFS = {"miccai2022": "/root/reference/MICCAI-2022", "mia2023": "/root/reference/MIA 2023/stage2_unimodal_student"}


def stub_torchvision():
    class _T:
        def __init__(self, *a, **k): pass
        def __call__(self, x): return x

    class _Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    for name in ("RandomHorizontalFlip", "RandomVerticalFlip", "RandomCrop", "ColorJitter", "ToTensor", "Normalize", "Resize"):
        setattr(tr, name, _T)
    tr.Compose = _Compose
    ds = types.ModuleType("torchvision.datasets")
    tv.transforms, tv.datasets = tr, ds
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.datasets": ds})


def load_ref(tag):
    """Import <ref>/data_loaders_MT.py under a private name, with <ref> first on sys.path for its `from utils import ...`."""
    root = REFS[tag]
    for k in [k for k in sys.modules if k in ("utils", "data_loaders_MT", "options")]:
        del sys.modules[k]
    sys.path.insert(0, root)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        spec = importlib.util.spec_from_file_location("ref_loader_" + tag, os.path.join(root, "data_loaders_MT.py"))
        mod = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
    finally:
        os.chdir(cwd)
        sys.path.remove(root)
    return mod


def main():
    from PIL import Image
    MG.install_shims()
    stub_torchvision()
    n = 30
    labels = np.random.RandomState(5).permutation(np.repeat(np.arange(3), n // 3)).astype(np.float64)
    tmp = tempfile.mkdtemp()
    paths = []
    for i in range(n):
        p = os.path.join(tmp, "tile_%02d.png" % i)
        Image.fromarray(np.full((8, 8, 3), i, dtype=np.uint8)).save(p)
        paths.append(p)
    order = np.random.RandomState(6).permutation(n)[:12]
    rec = dict(labels=labels.astype(np.int64), order=order, n=n)
    cases = []
    for pos_mode in ("exact", "relax", "multi_pos"):
        for K in (8, 50):                      # below / above the 20 other-class rows: replace False / True (:243)
            cases.append(("miccai2022", pos_mode, "diff_class", 6, K))
    for neg_mode, K in (("all_others", 8), ("all_others", 40), ("diff_class", 8), ("diff_class", 50)):
        cases.append(("mia2023", "multi_pos", neg_mode, 6, K))
    cases.append(("mia2023", "relax", "all_others", 6, 8))
    mods = {tag: load_ref(tag) for tag in REFS}
    names = []
    for ci, (tag, pos_mode, neg_mode, P, K) in enumerate(cases):
        opt = SimpleNamespace(nce_p=P, nce_k=K, pos_mode=pos_mode, neg_mode=neg_mode, distill="crd", task="grad", label_dim=3,
                              input_size_path=8, dataroot="", mode="pathomic")
        data = {"train": dict(x_path=list(paths), x_omic=np.zeros((n, 4), np.float32), e=np.zeros(n), t=np.zeros(n),
                              g=labels.copy())}
        with contextlib.redirect_stdout(io.StringIO()):
            ds = mods[tag].Pathomic_InstanceSample(opt, data, split="train", mode="pathomic")
        seed = 100 + ci
        np.random.seed(seed)
        rows = []
        for index in order:
            item = ds[int(index)]
            assert int(item[6]) == int(index)
            rows.append(np.asarray(item[7]).astype(np.int64))
        name = f"c{ci}"
        names.append(name)
        rec.update({name + "_rows": np.stack(rows), name + "_seed": seed, name + "_P": P, name + "_K": K,
                    name + "_pos_mode": pos_mode, name + "_neg_mode": neg_mode, name + "_ref": tag})
    rec["cases"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, "sampler_draws.npz"), **rec)
    print("wrote sampler_draws.npz:", len(cases), "cases")


if __name__ == "__main__":
    main()
