"""Parity of the loss kernels on the GPU against golden vectors from the reference: DistillKL (KD_loss.py),
CRDLoss with DC-Distill selection (CL_utils/CRD_loss.py + memory_new.py), GK-Refine (AEKD_loss)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_distill_kl(golden_dir):
    import multimodal_learning_amd as m
    from tests.gpu_util import assert_close
    g = np.load(os.path.join(golden_dir, "modules_b4_h64.npz"))
    for T in (1, 4):
        ys = torch.as_tensor(g["kl_ys"]).cuda().requires_grad_(True)
        kl = m.DistillKL(float(T))(ys, torch.as_tensor(g["kl_yt"]).cuda())
        gr, = torch.autograd.grad(kl, ys)
        assert kl.dim() == 0
        assert_close(g[f"kl_T{T}"], kl, 1e-6, 1e-5, "kl"); assert_close(g[f"kl_T{T}_g"], gr, 1e-6, 1e-5, "kl grad")


@pytest.mark.parametrize("mode", ["mid", "hard"])
def test_crd_loss_golden(golden_dir, mode):
    """Two consecutive calls: Z is set on the first and frozen on the second; banks are momentum-updated."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.losses import CRDState
    from oracle.step import default_opt
    from tests.gpu_util import assert_close
    g = np.load(os.path.join(golden_dir, f"crd_{mode}.npz"))
    crd = m.CRDLoss(default_opt(select_pos_mode=mode), 1024)
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11))
    st = CRDState(1024, seed=int(g["bank_seed"]))
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    crd = crd.cuda()
    crd.contrast.verbose = False
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).cuda().requires_grad_(True)
        ranks = g["ranks"][it] if mode == "mid" else None
        loss = crd(0.1, f_s, torch.as_tensor(g[f"f_t{it}"]).cuda(), torch.as_tensor(g[f"index{it}"]).cuda(),
                   torch.as_tensor(g[f"sidx{it}"]).cuda(), ranks=ranks)
        assert loss.dim() == 0
        gs = torch.autograd.grad(loss, [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight,
                                        crd.embed_s.linear.bias])
        assert_close(g[f"loss{it}"], loss, 1e-4, 1e-5, "crd loss")
        assert_close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3, "d f_s"); assert_close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3, "d W_s")
        assert_close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3, "d W_t"); assert_close(g[f"g_bs{it}"], gs[3], 1e-6, 1e-3, "d b_s")
        assert_close(g[f"params{it}"], crd.contrast.params, 1e-2, 1e-4, "params (Z)")
        idx = torch.as_tensor(g[f"index{it}"]).cuda()
        assert_close(g[f"bank_v1_rows{it}"], crd.contrast.memory_v1[idx], 1e-6, 0, "bank v1 rows")
        assert_close(g[f"bank_v2_rows{it}"], crd.contrast.memory_v2[idx], 1e-6, 0, "bank v2 rows")


@pytest.mark.parametrize("mode", ["mid", "hard"])
def test_crd_loss_sample_kd_golden(golden_dir, mode):
    """VERDICT r04 missing 1: `--sample_KD True` INSIDE CRDLoss (CRD_loss.py:148-149 -> ContrastLoss_v2 :246-250): forward
    returns the [B] per-sample losses s_loss + t_loss.  Reference golden, two calls (Z set, then frozen)."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.losses import CRDState
    from oracle.step import default_opt
    from tests.gpu_util import assert_close
    g = np.load(os.path.join(golden_dir, "crd_samplekd.npz"))
    opt = default_opt(select_pos_mode=mode, sample_KD="True", nce_p=int(g["nce_p"]), nce_k=int(g["nce_k"]),
                      nce_p2=int(g["nce_p2"]), nce_k2=int(g["nce_k2"]))
    crd = m.CRDLoss(opt, int(g["n_data"]))
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11))
    st = CRDState(int(g["n_data"]), seed=int(g["bank_seed"]))
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    crd = crd.cuda()
    crd.contrast.verbose = False
    for it in range(2):
        t = f"{mode}{it}"
        f_s = torch.as_tensor(g[f"f_s_{t}"]).cuda().requires_grad_(True)
        ranks = g[f"ranks_{mode}"][it] if mode == "mid" else None
        rows = crd(0.1, f_s, torch.as_tensor(g[f"f_t_{t}"]).cuda(), torch.as_tensor(g[f"index_{t}"]).cuda(),
                   torch.as_tensor(g[f"sidx_{t}"]).cuda(), ranks=ranks)
        assert tuple(rows.shape) == (f_s.shape[0],)
        gs = torch.autograd.grad((rows * torch.as_tensor(g[f"w_{t}"]).cuda()).sum(),
                                 [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight, crd.embed_s.linear.bias])
        assert_close(g[f"rows_{t}"], rows, 1e-3, 1e-5, "per-sample crd losses")
        assert_close(g[f"g_fs_{t}"], gs[0], 1e-5, 1e-3, "d f_s"); assert_close(g[f"g_ws_{t}"], gs[1], 1e-5, 1e-3, "d W_s")
        assert_close(g[f"g_wt_{t}"], gs[2], 1e-5, 1e-3, "d W_t"); assert_close(g[f"g_bs_{t}"], gs[3], 1e-5, 1e-3, "d b_s")
        assert_close(g[f"params_{t}"], crd.contrast.params, 1e-2, 1e-4, "params (Z)")
        idx = torch.as_tensor(g[f"index_{t}"]).cuda()
        assert_close(g[f"bank_v1_rows_{t}"], crd.contrast.memory_v1[idx], 1e-6, 0, "bank v1 rows")
        assert_close(g[f"bank_v2_rows_{t}"], crd.contrast.memory_v2[idx], 1e-6, 0, "bank v2 rows")


@pytest.mark.parametrize("mode", ["mid", "hard"])
def test_contrast_memory_v3_forward_standalone_vs_reference_golden(golden_dir, mode):
    """Rows a8 as NAMED (VERDICT r02 missing 5): ContrastMemory_v3.forward(epoch, v1, v2, y, idx, select_pos_mode) ->
    (out_v1, out_v2) [B, P2+K2, 1] against the reference class run standalone - two calls (Z set, then frozen; the
    second call scores against the momentum-updated bank), outputs, gradients w.r.t. v1 / v2 taken AFTER the in-call
    bank update (the backward must use the pre-update rows), Z, updated rows.  In `mid` mode one call draws its rank
    list from numpy's global RNG like the reference (same seed -> same list), the other gets it passed in."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.CL_utils.memory_new import ContrastMemory_v3
    from oracle.losses import CRDState
    from tests.gpu_util import assert_close
    g = np.load(os.path.join(golden_dir, "crd_forward.npz"))
    mem = ContrastMemory_v3(128, int(g["n_data"]), int(g["P"]), int(g["K"]), float(g["T"]), float(g["momentum"]), True,
                            int(g["P2"]), "True", int(g["K2"]))
    st = CRDState(int(g["n_data"]), seed=int(g["bank_seed"]))
    mem.memory_v1.copy_(st.memory_v1); mem.memory_v2.copy_(st.memory_v2)
    mem = mem.cuda(); mem.verbose = False
    np.random.seed(11)                       # the generator's seed: the first `mid` call draws the reference's list itself
    for it in range(2):
        t = f"{mode}{it}"
        v1 = torch.as_tensor(g[f"v1_{t}"]).cuda().requires_grad_(True)
        v2 = torch.as_tensor(g[f"v2_{t}"]).cuda().requires_grad_(True)
        y, idx = torch.as_tensor(g[f"y_{t}"]).cuda(), torch.as_tensor(g[f"idx_{t}"]).cuda()
        ranks = g[f"ranks_{mode}"][it] if (mode == "mid" and it == 1) else None
        o1, o2 = mem(0.1, v1, v2, y, idx, select_pos_mode=mode, ranks=ranks)
        assert o1.shape == o2.shape == (v1.shape[0], int(g["P2"]) + int(g["K2"]), 1)
        gv1, gv2 = torch.autograd.grad((o1 * torch.as_tensor(g[f"w1_{t}"]).cuda()).sum()
                                       + (o2 * torch.as_tensor(g[f"w2_{t}"]).cuda()).sum(), [v1, v2])
        assert_close(g[f"out1_{t}"], o1, 1e-9, 1e-4, "out_v1"); assert_close(g[f"out2_{t}"], o2, 1e-9, 1e-4, "out_v2")
        assert_close(g[f"gv1_{t}"], gv1, 1e-6, 1e-3, "d v1"); assert_close(g[f"gv2_{t}"], gv2, 1e-6, 1e-3, "d v2")
        assert_close(g[f"params_{t}"], mem.params, 1e-2, 1e-4, "params (Z)")
        assert_close(g[f"rows1_{t}"], mem.memory_v1[y], 1e-6, 0, "bank v1 rows")
        assert_close(g[f"rows2_{t}"], mem.memory_v2[y], 1e-6, 0, "bank v2 rows")


def test_contrast_loss_v2_forward_standalone_vs_reference_golden(golden_dir):
    """Row a9 as named: ContrastLoss_v2.forward(x, P), both sample_KD branches (CRD_loss.py:240-250), loss + d x."""
    from multimodal_learning_amd.CL_utils.CRD_loss import ContrastLoss_v2
    from tests.gpu_util import assert_close
    g = np.load(os.path.join(golden_dir, "crd_forward.npz"))
    for kd in ("False", "True"):
        x = torch.as_tensor(g["cl_x"]).cuda().requires_grad_(True)
        loss = ContrastLoss_v2(int(g["n_data"]), kd)(x, int(g["P2"]))
        assert tuple(loss.shape) == tuple(g[f"cl_loss_{kd}"].shape)
        (gx,) = torch.autograd.grad((loss * torch.as_tensor(g[f"cl_w_{kd}"]).cuda()).sum(), [x])
        assert_close(g[f"cl_loss_{kd}"], loss, 1e-5, 1e-5, "ContrastLoss_v2 " + kd)
        assert_close(g[f"cl_gx_{kd}"], gx, 1e-6, 1e-4, "d x " + kd)


def test_crd_select_kernel_bit_exact():
    """Integer work is bit-exact: given the SAME discrepancy values, ph_crd_select returns exactly the columns
    torch.sort-based selection (memory_new.py:303-345) returns - 'hard', 'mid' ranks, negatives on/off."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    L = lib()
    g = torch.Generator().manual_seed(0)
    for (B, P, K, P2, K2) in ((16, 300, 700, 20, 512), (3, 100, 37, 5, 37), (1, 128, 4096, 6, 1000)):
        PK = P + K
        diff = torch.randn(B, PK, generator=g)           # continuous values: no ties
        out1 = torch.rand(B, PK, generator=g); out2 = torch.rand(B, PK, generator=g)
        for ranks in (None, np.random.RandomState(1).choice(np.arange(min(30, P - P2), P), P2, replace=False)):
            for select_neg in (1, 0):
                k2 = K2 if select_neg else K
                idx_pos = torch.sort(diff[:, :P], dim=1, descending=True)[1]
                sel_pos = (idx_pos[:, :P2] if ranks is None else idx_pos[:, torch.as_tensor(ranks)]).clone()
                sel_pos[:, 0] = 0
                if select_neg:
                    sel_neg = P + torch.sort(diff[:, P:], dim=1, descending=False)[1][:, :k2]
                else:
                    sel_neg = P + torch.arange(K).expand(B, K)
                ref = torch.cat([sel_pos, sel_neg], 1)
                sel = torch.empty(B, P2 + k2, dtype=torch.int32, device="cuda")
                xs = torch.empty(B, P2 + k2, device="cuda"); xt = torch.empty_like(xs)
                r = None if ranks is None else torch.as_tensor(ranks, dtype=torch.int32).cuda()
                d, o1, o2 = diff.cuda(), out1.cuda(), out2.cuda()
                check(L.ph_crd_select(ptr(d), ptr(o1), ptr(o2), ptr(r), ptr(sel), ptr(xs), ptr(xt), B, P, K, P2, k2,
                                      select_neg, 1, stream()), "select")
                assert torch.equal(sel.cpu().long(), ref)
                assert torch.equal(xs.cpu(), torch.gather(out1, 1, ref)) and torch.equal(xt.cpu(), torch.gather(out2, 1, ref))


def test_crd_selection_end_to_end_overlap():
    """End to end the discrepancies come from different fp32 summation orders, so ranks whose discrepancies
    are closer than ~1e-7 may swap (exact-integer behaviour is pinned by test_crd_select_kernel_bit_exact):
    >= 90 % of the rank-picked positives identical, negative sets overlap >= 99 %, loss agrees to 1e-4."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.losses import CRDState, crd_loss
    from oracle.step import default_opt
    opt = default_opt()
    n_data, B = 4096, 16
    st = CRDState(n_data, seed=3, embed_s=W.make_state_dict(W.embed_shapes(), 1), embed_t=W.make_state_dict(W.embed_shapes(), 2))
    crd = m.CRDLoss(opt, n_data)
    crd.embed_s.load_state_dict(st.embed_s); crd.embed_t.load_state_dict(st.embed_t)
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    crd = crd.cuda(); crd.contrast.verbose = False
    g = torch.Generator().manual_seed(0)
    f_s = torch.randn(B, 128, generator=g).relu(); f_t = torch.randn(B, 128, generator=g).relu()
    index = torch.randperm(n_data, generator=g)[:B]
    sidx = torch.randint(0, n_data, (B, 1000), generator=g); sidx[:, 0] = index
    ranks = np.random.RandomState(0).choice(np.arange(30, 100), 20, replace=False)
    ref, aux = crd_loss(st, f_s, f_t, index, sidx, 20, 512, "mid", ranks, return_aux=True)
    loss = crd(0.0, f_s.cuda(), f_t.cuda(), index.cuda(), sidx.cuda(), ranks=ranks)
    sel = crd.contrast.last["sel"].cpu().long()
    same_pos = (sel[:, :20] == aux["sel_pos"]).float().mean().item()
    overlap = np.mean([len(set(sel[b, 20:].tolist()) & set(aux["sel_neg"][b].tolist())) / 512.0 for b in range(B)])
    print(f"\nCRD selection: positives identical {same_pos:.4f}, negative-set overlap {overlap:.4f}")
    assert same_pos >= 0.90 and overlap >= 0.99
    assert abs(loss.item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))


def test_aekd_loss_vs_oracle():
    import multimodal_learning_amd as m
    from oracle.losses import aekd_loss
    from oracle.step import default_opt
    opt = default_opt()
    g = torch.Generator().manual_seed(4)
    B = 16
    feat_c = torch.randn(B, 128, generator=g)
    ws = [torch.randn(128, generator=g) for _ in range(5)]

    def mk(feat):
        return [((feat * w).sum(1) ** 2).mean() * (0.1 + i) for i, w in enumerate(ws)]
    f1 = feat_c.clone().requires_grad_(True)
    l1 = mk(f1)
    sc_ref, tot_ref = aekd_loss(l1[4], f1, l1[:4], True)
    f2 = feat_c.clone().cuda().requires_grad_(True)
    ws = [w.cuda() for w in ws]
    l2 = mk(f2)
    sc, tot = m.AEKD_loss(opt, None, l2[4], f2, l2[:4])
    assert sc.shape == (5,) and not sc.requires_grad
    assert torch.allclose(sc.cpu(), sc_ref, rtol=1e-4, atol=1e-5)
    assert abs(tot.item() - tot_ref.item()) < 1e-4 * abs(tot_ref.item())
    gt, = torch.autograd.grad(tot, f2)
    gr, = torch.autograd.grad(tot_ref, f1)
    assert torch.allclose(gt.cpu(), gr, rtol=1e-3, atol=1e-6)


def test_mia2022_crd_v3_golden(golden_dir):
    """vanilla K+1 CRD bank with a per-sample-weighted loss vs the MIA-2022 reference."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.CL_utils import CRD_criterion_v3 as V3
    from oracle import weights as W
    from oracle.variants import CRDv3State
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "mia2022_crd_v3.npz"))
    opt = m.stage2_opt(nce_k=int(g["K"]))
    crd = V3.CRDLoss(opt, int(g["n_data"]))
    assert set(crd.state_dict()) == {"embed_s.linear.weight", "embed_s.linear.bias", "embed_t.linear.weight",
                                     "embed_t.linear.bias", "contrast.params", "contrast.memory_v1", "contrast.memory_v2"}
    assert crd.contrast.params.numel() == 5
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 30))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 31))
    st = CRDv3State(int(g["n_data"]), K=int(g["K"]), seed=int(g["bank_seed"]))
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    crd = crd.cuda(); crd.contrast.verbose = False
    R = Report("MIA-2022 CRD_criterion_v3 vs reference golden")
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).cuda().requires_grad_(True)
        loss = crd(float(g[f"w{it}"]), f_s, torch.as_tensor(g[f"f_t{it}"]).cuda(), torch.as_tensor(g[f"index{it}"]).cuda(),
                   torch.as_tensor(g[f"sidx{it}"]).cuda())
        assert tuple(loss.shape) == (1,)
        gs = torch.autograd.grad(loss.sum(), [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight])
        R.close(g[f"loss{it}"], loss, 1e-4, 1e-5, f"loss call {it}")
        R.close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3, f"d f_s call {it}"); R.close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3, f"d W_s call {it}")
        R.close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3, f"d W_t call {it}")
        R.close(g[f"params{it}"], crd.contrast.params, 1e-2, 1e-4, f"params/Z call {it}")
        R.close(g[f"bank_v1_rows{it}"], crd.contrast.memory_v1[torch.as_tensor(g[f"index{it}"]).cuda()], 1e-6, 0, f"bank rows call {it}")
    R.finish()


def test_mia2022_momentum_gk_golden(golden_dir):
    import multimodal_learning_amd as m
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "mia2022_momentum_gk.npz"))
    ws = torch.as_tensor(g["ws"]).cuda()
    R = Report("MIA-2022 momentum_AEKD_loss vs reference golden")
    for name, gth, th in (("plain", "False", 0.0), ("thresh", "True", 0.25)):
        opt = m.stage2_opt()
        opt.grads_thresh, opt.thresh, opt.grads_m = gth, th, 0.9
        mo = None
        for it in range(3):
            feat = (torch.as_tensor(g["feat"]).cuda() * (1 + 0.1 * it)).clone().requires_grad_(True)
            losses = [((feat * w).sum(1) ** 2).mean() * (0.1 + i) + (feat ** 2).mean() * (i % 2) for i, w in enumerate(ws)]
            mo, total = m.momentum_AEKD_loss(opt, None, losses[4], feat, losses[:4], mo)
            R.close(g[f"{name}_scale{it}"], mo, 1e-4, 1e-4, f"{name} mo_scale it {it}")
            R.close(g[f"{name}_total{it}"], total, 1e-4, 1e-4, f"{name} total it {it}")
    R.finish()


def test_mia2023_crd_v10_golden(golden_dir):
    """class-masked full-bank KNN positives on the GPU + similarity-weighted, per-sample-weighted NCE
    vs the MIA-2023 reference (which does the bank scan with sklearn on the host)."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.CL_utils import CRD_criterion_v10 as V10
    from oracle import weights as W
    from oracle.variants import CRDv10State
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "mia2023_crd_v10.npz"))
    labels = torch.as_tensor(g["labels"])
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    opt = m.stage2_opt(nce_k=int(g["K"]), nce_p=int(g["num_pos"]))
    crd = V10.CRDLoss(opt, int(g["n_data"]), class_idx)
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 50))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 51))
    st = CRDv10State(int(g["n_data"]), labels, K=int(g["K"]), seed=int(g["bank_seed"]))
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    crd = crd.cuda(); crd.contrast.verbose = False
    R = Report("MIA-2023 CRD_criterion_v10 (neighbors) vs reference golden")
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).cuda().requires_grad_(True)
        loss, sl = crd(torch.as_tensor(g[f"w{it}"]).cuda(), f_s, torch.as_tensor(g[f"f_t{it}"]).cuda(),
                       torch.as_tensor(g[f"grade{it}"]).cuda(), torch.as_tensor(g[f"index{it}"]).cuda(),
                       torch.as_tensor(g[f"sidx{it}"]).cuda())
        gs = torch.autograd.grad(loss, [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight])
        R.close(g[f"loss{it}"], loss, 1e-4, 1e-5, f"loss call {it}")
        R.close(g[f"sample_loss{it}"], sl, 1e-3, 1e-5, f"sample_loss call {it}")
        R.close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3, f"d f_s call {it}"); R.close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3, f"d W_s call {it}")
        R.close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3, f"d W_t call {it}")
        R.close(g[f"params{it}"], crd.contrast.params, 1e-2, 1e-4, f"params/Z call {it}")
        R.close(g[f"bank_v1_rows{it}"], crd.contrast.memory_v1[torch.as_tensor(g[f"index{it}"]).cuda()], 1e-6, 0, f"bank rows call {it}")
    R.finish()


@pytest.mark.parametrize("n_data,K,B,form", [(2048, 4096, 8, "scan"), (2048, 4096, 8, "gathered"), (1000, 1000, 5, "scan"),
                                             (65536, 65536, 6, "scan"), (65536, 65536, 6, "gathered")])
def test_mia2023_crd_v10_with_as_many_negatives_as_bank_rows(n_data, K, B, form, monkeypatch):
    """BASELINE configs[4] read as nce_k = 65536 negatives per query (SURVEY 8-e assumption (i), VERDICT r04 missing 2): with
    nce_k at or above the number of bank rows CRDLoss sums the negatives' terms over the whole bank weighted by multiplicity
    (bank-scan form: memory_new._crd_core_scan) - against the CPU oracle's gathered evaluation of CRD_criterion_v10.py (pinned on
    the reference's golden at small K), two calls (first-call Z, momentum update), loss / per-sample losses / gradients / Z / bank
    rows; the gathered kernels (PH_CRD_SCAN=0) must agree on the same inputs, whatever the list length."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.CL_utils import CRD_criterion_v10 as V10
    from oracle import weights as W
    from oracle.variants import CRDv10State, crd_v10_loss
    from tests.gpu_util import Report
    monkeypatch.setenv("PH_CRD_SCAN", "1" if form == "scan" else "0")
    NP = 6
    g = torch.Generator().manual_seed(n_data + K + B)
    labels = torch.arange(n_data) % 3
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    opt = m.stage2_opt(nce_k=K, nce_p=NP)
    crd = V10.CRDLoss(opt, n_data, class_idx)
    es, et = W.make_state_dict(W.embed_shapes(), 60), W.make_state_dict(W.embed_shapes(), 61)
    crd.embed_s.load_state_dict(es); crd.embed_t.load_state_dict(et)
    st = CRDv10State(n_data, labels, K=K, seed=77, embed_s=es, embed_t=et)
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    crd = crd.cuda(); crd.contrast.verbose = False
    R = Report("MIA-2023 CRD v10, nce_k %d over %d bank rows (%s form) vs oracle" % (K, n_data, form))
    for it in range(2):
        index = torch.randperm(n_data, generator=g)[:B]
        sidx = torch.randint(0, n_data, (B, K + 1), generator=g); sidx[:, 0] = index
        grade = labels[index]
        f_s = torch.randn(B, 128, generator=g); f_t = torch.randn(B, 128, generator=g)
        w = torch.rand(B, generator=g) + 0.5
        fo = f_s.clone().requires_grad_(True)
        lo, slo, _ = crd_v10_loss(st, w, fo, f_t, grade, index, sidx, NP)
        go, = torch.autograd.grad(lo, fo)
        fg = f_s.cuda().requires_grad_(True)
        loss, sl = crd(w.cuda(), fg, f_t.cuda(), grade.cuda(), index.cuda(), sidx.cuda())
        gg = torch.autograd.grad(loss, [fg, crd.embed_t.linear.weight])
        assert (crd.contrast._scan_neg is not None) == (form == "scan")
        R.close(lo.detach().numpy(), loss, 1e-4, 1e-5, f"loss call {it}")
        R.close(slo.detach().numpy(), sl, 1e-3, 1e-5, f"sample_loss call {it}")
        R.close(go.numpy(), gg[0], 1e-6, 2e-3, f"d f_s call {it}")
        assert torch.isfinite(gg[1]).all()
        R.close(st.params.numpy(), crd.contrast.params, 1e-2, 1e-4, f"params/Z call {it}")
        R.close(st.memory_v1[index].numpy(), crd.contrast.memory_v1[index.cuda()], 1e-6, 0, f"bank rows call {it}")
    R.finish()


def test_mia2023_crd_v10_centers_golden(golden_dir):
    """`--pos_extra centers --nce_p 2` (CRD_criterion_v10.py:81-101): class-mean positives / negatives vs the reference's
    CRDLoss, two calls (Z set on the first, the centres recomputed from the updated bank on the second); state_dict keeps
    the reference's [n_data, 128] bank shape although the centres live behind the bank rows; nce_p > 2 raises."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.CL_utils import CRD_criterion_v10 as V10
    from oracle import weights as W
    from oracle.variants import CRDv10State
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "mia2023_crd_v10_centers.npz"))
    labels = torch.as_tensor(g["labels"])
    n_data = int(g["n_data"])
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    opt = m.stage2_opt(nce_k=int(g["K"]), nce_p=int(g["num_pos"]), pos_extra="centers")
    crd = V10.CRDLoss(opt, n_data, class_idx)
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 52))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 53))
    st = CRDv10State(n_data, labels, K=int(g["K"]), seed=int(g["bank_seed"]))
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    crd = crd.cuda(); crd.contrast.verbose = False
    R = Report("MIA-2023 CRD_criterion_v10 (centers, nce_p 2) vs reference golden")
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).cuda().requires_grad_(True)
        bank_before = crd.contrast.memory_v1.detach().clone()
        loss, sl = crd(torch.as_tensor(g[f"w{it}"]).cuda(), f_s, torch.as_tensor(g[f"f_t{it}"]).cuda(),
                       torch.as_tensor(g[f"grade{it}"]).cuda(), torch.as_tensor(g[f"index{it}"]).cuda(),
                       torch.as_tensor(g[f"sidx{it}"]).cuda())
        gs = torch.autograd.grad(loss, [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight])
        R.close(g[f"loss{it}"], loss, 1e-4, 1e-5, f"loss call {it}")
        R.close(g[f"sample_loss{it}"], sl, 1e-3, 1e-5, f"sample_loss call {it}")
        R.close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3, f"d f_s call {it}"); R.close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3, f"d W_s call {it}")
        R.close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3, f"d W_t call {it}")
        R.close(g[f"params{it}"], crd.contrast.params, 1e-2, 1e-4, f"params/Z call {it}")
        ix = torch.as_tensor(g[f"index{it}"]).cuda()
        R.close(g[f"bank_v1_rows{it}"], crd.contrast.memory_v1[ix], 1e-6, 0, f"bank-1 rows call {it}")
        R.close(g[f"bank_v2_rows{it}"], crd.contrast.memory_v2[ix], 1e-6, 0, f"bank-2 rows call {it}")
        # the centre rows behind the bank are the class means of the PRE-update bank
        ext = crd.contrast._ext_memory_v1
        for c in range(3):
            R.close(bank_before[torch.as_tensor(class_idx[c]).cuda()].double().mean(0), ext[n_data + c], 2e-7, 1e-6, f"centre {c} call {it}")
    R.finish()
    sd = crd.state_dict()
    assert tuple(sd["contrast.memory_v1"].shape) == (n_data, 128) and tuple(sd["contrast.memory_v2"].shape) == (n_data, 128)
    crd2 = V10.CRDLoss(opt, n_data, class_idx).cuda()
    crd2.load_state_dict(sd)
    assert torch.equal(crd2.contrast.memory_v1, crd.contrast.memory_v1)
    with pytest.raises(NotImplementedError):
        V10.CRDLoss(m.stage2_opt(nce_k=16, nce_p=3, pos_extra="centers"), n_data, class_idx)
    with pytest.raises(NotImplementedError):
        V10.CRDLoss(m.stage2_opt(nce_k=16, nce_p=3, pos_extra="prototypes"), n_data, class_idx)


@pytest.mark.parametrize("n,B,NP", [(65536, 8, 6), (65536, 64, 6), (5000, 100, 8), (77, 3, 2)])
def test_mia2023_bank_topk_bit_exact(n, B, NP):
    """The KNN indices are integer work: identical to torch.sort of the class-masked cosine on the same bank.  (65536 rows =
    BASELINE config 5's bank; 64 queries = the benchmark batch; 100 queries = two passes of <= 64; ragged last tile.)"""
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    import torch.nn.functional as F
    L = lib()
    g = torch.Generator().manual_seed(2)
    mem1 = torch.rand(n, 128, generator=g) - 0.5; mem2 = torch.rand(n, 128, generator=g) - 0.5
    labels = torch.randint(0, 3, (n,), generator=g).int()
    idx = torch.randint(0, n, (B, 5), generator=g)
    bl = labels[idx[:, 0]].long()
    d = lambda t: t.cuda()
    nb1 = torch.empty(B, NP, dtype=torch.int64, device="cuda"); nb2 = torch.empty_like(nb1)
    s1 = torch.empty(B, NP, device="cuda"); s2 = torch.empty_like(s1)
    m1, m2, lb, ix, blc = d(mem1), d(mem2), d(labels), d(idx), d(bl)     # keep the device tensors alive across the launch
    ws = torch.empty(L.ph_crd_bank_topk_workspace_bytes(B, n), dtype=torch.uint8, device="cuda")
    check(L.ph_crd_bank_topk(ptr(m1), ptr(m2), ptr(lb), ptr(ix), 5, ptr(blc), B, n, NP, 128,
                             ptr(nb1), ptr(nb2), ptr(s1), ptr(s2), ptr(ws), stream()), "topk")
    for mem, nb, s in ((mem1, nb1, s1), (mem2, nb2, s2)):
        sim = (labels.view(1, -1) == bl.view(-1, 1)).float() * (F.normalize(mem[idx[:, 0]], dim=1) @ F.normalize(mem, dim=1).T)
        srt = torch.sort(sim, descending=True, dim=-1)
        assert torch.equal(nb.cpu(), srt[1][:, :NP])
        assert torch.allclose(s.cpu(), srt[0][:, :NP], atol=1e-6)


def test_mia2023_bank_topk_rare_class_and_sorted_bank():
    """Cases the seeded selection must survive unchanged: a class with fewer members than num_pos (the masked zeros fill the
    list in row order - torch.sort(stable=True)), and a bank sorted by class (the sample sees the classes in blocks)."""
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    import torch.nn.functional as F
    L = lib()
    g = torch.Generator().manual_seed(5)
    n, B, NP = 20000, 16, 8
    for case in ("rare", "sorted"):
        mem1 = torch.rand(n, 128, generator=g) - 0.5; mem2 = torch.rand(n, 128, generator=g) - 0.5
        if case == "rare":
            labels = torch.randint(0, 2, (n,), generator=g).int()
            rare = torch.tensor([17, 4000, 9999, 19998])
            labels[rare] = 2
            idx = torch.randint(0, n, (B, 5), generator=g)
            idx[:4, 0] = rare
        else:
            labels = (torch.arange(n) * 3 // n).int()
            idx = torch.randint(0, n, (B, 5), generator=g)
        bl = labels[idx[:, 0]].long()
        nb1 = torch.empty(B, NP, dtype=torch.int64, device="cuda"); nb2 = torch.empty_like(nb1)
        s1 = torch.empty(B, NP, device="cuda"); s2 = torch.empty_like(s1)
        m1, m2, lb, ix, blc = mem1.cuda(), mem2.cuda(), labels.cuda(), idx.cuda(), bl.cuda()
        ws = torch.empty(L.ph_crd_bank_topk_workspace_bytes(B, n), dtype=torch.uint8, device="cuda")
        check(L.ph_crd_bank_topk(ptr(m1), ptr(m2), ptr(lb), ptr(ix), 5, ptr(blc), B, n, NP, 128,
                                 ptr(nb1), ptr(nb2), ptr(s1), ptr(s2), ptr(ws), stream()), "topk")
        for mem, nb, s in ((mem1, nb1, s1), (mem2, nb2, s2)):
            sim = (labels.view(1, -1) == bl.view(-1, 1)).float() * (F.normalize(mem[idx[:, 0]], dim=1) @ F.normalize(mem, dim=1).T)
            sim = sim + 0.0      # (-0.0 of a masked negative similarity sorts like +0.0)
            srt = torch.sort(sim, descending=True, dim=-1, stable=True)
            assert torch.equal(nb.cpu(), srt[1][:, :NP]), case
            assert torch.allclose(s.cpu(), srt[0][:, :NP], atol=1e-6), case


def test_mia2023_bank_topk_is_independent_of_batching_and_repeatable():
    """Size-independent properties at BASELINE config 5's bank (65 536 rows): a query's neighbours do not depend on which other
    queries share the call (64 at once == 2 x 32 == 64 x 1 for a few of them: different 32-query blocks, different thresholds from
    the sample pass, different list layouts - the same keys), and two calls give bitwise the same result."""
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    L = lib()
    g = torch.Generator().manual_seed(9)
    n, B, NP = 65536, 64, 6
    m1 = (torch.rand(n, 128, generator=g) - 0.5).cuda(); m2 = (torch.rand(n, 128, generator=g) - 0.5).cuda()
    lb = torch.randint(0, 3, (n,), generator=g).int().cuda()
    ix = torch.randint(0, n, (B, 5), generator=g).cuda()
    bl = lb[ix[:, 0]].long()

    def call(sl):
        b = sl.stop - sl.start
        nb1 = torch.empty(b, NP, dtype=torch.int64, device="cuda"); nb2 = torch.empty_like(nb1)
        s1 = torch.empty(b, NP, device="cuda"); s2 = torch.empty_like(s1)
        ws = torch.empty(L.ph_crd_bank_topk_workspace_bytes(b, n), dtype=torch.uint8, device="cuda")
        ixs, bls = ix[sl].contiguous(), bl[sl].contiguous()
        check(L.ph_crd_bank_topk(ptr(m1), ptr(m2), ptr(lb), ptr(ixs), 5, ptr(bls), b, n, NP, 128, ptr(nb1), ptr(nb2), ptr(s1), ptr(s2),
                                 ptr(ws), stream()), "topk")
        torch.cuda.synchronize()
        return nb1, nb2, s1, s2
    full = call(slice(0, B))
    again = call(slice(0, B))
    assert all(torch.equal(a, b) for a, b in zip(full, again))
    halves = [call(slice(0, 32)), call(slice(32, 64))]
    for k in range(4):
        assert torch.equal(full[k], torch.cat([h[k] for h in halves], 0))
    for q in (0, 31, 32, 63):
        one = call(slice(q, q + 1))
        for k in range(4):
            assert torch.equal(full[k][q:q + 1], one[k])


def test_mia2023_rows_golden(golden_dir):
    import multimodal_learning_amd as m
    from multimodal_learning_amd import mia2023
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "mia2023_rows.npz"))
    R = Report("MIA-2023 per-sample KL / sample weights / GK_refine_thresh vs reference golden")
    B = g["ys"].shape[0]
    for T in (1, 2):
        ys = torch.as_tensor(g["ys"]).cuda().requires_grad_(True)
        loss, sl = mia2023.DistillKL(float(T))(ys, torch.as_tensor(g["yt"]).cuda())
        gg, = torch.autograd.grad((sl * torch.arange(1, B + 1).float().cuda()).sum(), ys)
        R.close(g[f"kl_loss_T{T}"], loss, 1e-6, 1e-5, f"kl loss T={T}"); R.close(g[f"kl_rows_T{T}"], sl, 1e-6, 1e-5, f"kl rows T={T}")
        R.close(g[f"kl_g_T{T}"], gg, 1e-5, 1e-5, f"kl grad T={T}")
    ys, yt = torch.as_tensor(g["ys"]).cuda(), torch.as_tensor(g["yt"]).cuda()
    d = mia2023.assign_sample_weights(torch.softmax(ys, 1), torch.softmax(yt, 1), torch.as_tensor(g["grade"]).cuda(), 1, 1)
    R.close(g["discrep"], d, 1e-5, 0, "assign_sample_weights")
    ws = torch.as_tensor(g["ws"]).cuda()
    for name, use, th in (("thr", "True", 0.25), ("relu", "False", 0.2)):
        opt = m.stage2_opt(batch_size=B)
        opt.use_grads_thresh, opt.grads_thresh = use, th
        feat = torch.as_tensor(g["feat"]).cuda().clone().requires_grad_(True)
        rows = [((feat * w).sum(1) ** 2) * (0.1 + i) + (feat ** 2).mean(1) * (i % 2) for i, w in enumerate(ws)]
        scale, total = mia2023.GK_refine_thresh(opt, None, rows[4].mean(), feat, rows[:4])
        R.close(g[f"gk_{name}_scale"], scale, 1e-4, 1e-4, f"GK {name} scale"); R.close(g[f"gk_{name}_total"], total, 1e-4, 1e-4, f"GK {name} total")
    R.finish()


# ---------------------------------------------------------------------------------------------- row a16 (t-SVD)
def test_tsvd_adjacency_penalty_golden(golden_dir):
    """update_adj_tensor + Frobenius penalty vs the fixture produced by running "MIA 2022/train_test_tSVD.py"."""
    import multimodal_learning_amd as m
    g = np.load(os.path.join(golden_dir, "mia2022_tsvd.npz"))
    feats = [torch.tensor(g[f"feat{v}"], device="cuda", requires_grad=True) for v in range(4)]
    aux = [torch.tensor(g[f"aux{v}"], device="cuda") for v in range(4)]
    adj = m.tsvd.update_adj_tensor([None] * 4, feats)
    loss = m.tsvd.tsvd_penalty(adj, aux, float(g["mu"]))
    grads = torch.autograd.grad(loss, feats)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for v in range(4):
        assert np.allclose(adj[v].detach().cpu().numpy(), g[f"adj{v}"], atol=2e-6)
        ref = g[f"g_feat{v}"]
        assert np.abs(grads[v].cpu().numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-3)


@pytest.mark.parametrize("B,V,tau", [(16, 4, 0.05), (64, 4, 0.3), (24, 2, 0.1), (32, 8, 0.02), (8, 6, 0.5),
                                     (128, 4, 0.3), (65, 2, 0.1), (96, 6, 0.05), (128, 8, 0.02), (127, 4, 0.5)])
def test_tsvd_update_aux_vs_oracle(B, V, tau):
    """ph_tsvd_update_aux (B <= 64: Jacobi eigen-solver of X^H X in LDS; 64 < B <= 128, BASELINE config 4: one-sided
    Jacobi on the slice) vs the float64 numpy proximal operator of the oracle.  The
    algorithm itself is OUR reading of the absent `update_aux` (parity unpinned vs the reference)."""
    import multimodal_learning_amd as m
    from oracle import variants as OV
    rng = np.random.default_rng(B * 10 + V)
    feats = [torch.tensor(rng.standard_normal((B, 32)).clip(0), dtype=torch.float32) for _ in range(V)]
    adj = torch.stack(OV.update_adj_tensor(feats), dim=2)          # realistic input: row-normalised Gram matrices
    ref, tnn_ref = OV.update_aux(adj, tau)
    aux, tnn = m.tsvd.update_aux(adj.cuda(), tau)
    err = np.abs(aux.cpu().numpy().astype(np.float64) - ref).max()
    assert err <= 2e-5 * max(np.abs(ref).max(), 1.0), err
    assert abs(tnn.item() - tnn_ref) <= 1e-4 * max(abs(tnn_ref), 1.0)
    # idempotence-like property of a prox: tau = 0 reproduces the input
    aux0, _ = m.tsvd.update_aux(adj.cuda(), 0.0)
    assert np.abs(aux0.cpu().numpy() - adj.numpy()).max() <= 2e-5


def test_tsvd_state_step_runs_and_schedules_mu():
    import types
    import multimodal_learning_amd as m
    opt = types.SimpleNamespace(n_views=4, Lambda_global=0.05, mu=1e-5, pho=1.1, max_mu=1.0)
    st = m.tsvd.TSVDState(opt, 16, "cuda")
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(16, 32, generator=g).relu_().cuda().requires_grad_(True) for _ in range(4)]
    loss = st.step(feats)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(f.grad).all() for f in feats)
    assert abs(st.mu - 1.1e-5) < 1e-12 and st.tnn is not None


@pytest.mark.gpu
def test_stage1_orth_and_vanilla_crd_vs_reference_golden(golden_dir):
    """Row f-1 optional terms through the C-ABI: OrthLoss (row scaling kernels + fp32 GEMMs) and the stage-1 vanilla CRD
    criterion (two-layer heads over the fused CRD bank kernels) against the reference's own outputs."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.CL_utils.orthogonal_loss import OrthLoss
    from multimodal_learning_amd.CL_utils.CRD_criterion import CRDLoss
    from oracle.variants import CRDv3State
    from types import SimpleNamespace
    from tests.test_oracle_variants import _embed2_state
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "stage1_terms.npz"))
    R = Report("stage-1 optional terms vs REFERENCE golden")
    x1 = torch.as_tensor(g["orth_x1"]).cuda().requires_grad_(True); x2 = torch.as_tensor(g["orth_x2"]).cuda().requires_grad_(True)
    lo = OrthLoss()(x1, x2)
    g1, g2 = torch.autograd.grad(lo, [x1, x2])
    R.close(g["orth_loss"], lo, 1e-9, 1e-4, "orth loss"); R.close(g["orth_g1"], g1, 1e-9, 1e-3, "orth d/dx1")
    R.close(g["orth_g2"], g2, 1e-9, 1e-3, "orth d/dx2")
    opt = SimpleNamespace(s_dim=128, t_dim=128, feat_dim=128, nce_k=int(g["K"]), nce_t=0.07, nce_m=0.5, n_data=int(g["n_data"]))
    crd = CRDLoss(opt).cuda()
    crd.contrast.verbose = False
    crd.embed_s.load_state_dict(_embed2_state(70)); crd.embed_t.load_state_dict(_embed2_state(71))
    st = CRDv3State(opt.n_data, K=opt.nce_k, seed=80)
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).cuda().requires_grad_(True)
        idx = torch.as_tensor(g[f"index{it}"]).cuda()
        loss = crd(f_s, torch.as_tensor(g[f"f_t{it}"]).cuda(), idx, torch.as_tensor(g[f"sidx{it}"]).cuda())
        gs = torch.autograd.grad(loss.sum(), [f_s, crd.embed_s.linear[0].weight, crd.embed_s.linear[2].weight,
                                              crd.embed_t.linear[2].bias])
        R.close(g[f"loss{it}"], loss, 1e-4, 1e-5, f"CRD loss call {it}")
        R.close(g[f"g_fs{it}"], gs[0], 1e-6, 2e-3, f"d/df_s call {it}")
        R.close(g[f"g_w0{it}"], gs[1], 1e-6, 2e-3, f"d/d embed_s.linear.0.weight call {it}")
        R.close(g[f"g_w2{it}"], gs[2], 1e-6, 2e-3, f"d/d embed_s.linear.2.weight call {it}")
        R.close(g[f"g_tb2{it}"], gs[3], 1e-6, 2e-3, f"d/d embed_t.linear.2.bias call {it}")
        R.close(g[f"params{it}"], crd.contrast.params, 1e-2, 1e-5, "params / Z")
        R.close(g[f"bank_v1_rows{it}"], crd.contrast.memory_v1[idx], 1e-5, 0, f"bank v1 rows call {it}")
        R.close(g[f"bank_v2_rows{it}"], crd.contrast.memory_v2[idx], 1e-5, 0, f"bank v2 rows call {it}")
    R.finish()


@pytest.mark.gpu
def test_distiller_zoo_sp_and_feats_kl_vs_reference_golden(golden_dir):
    """Row f-4 (part): Similarity (SP) and feats_KL against the reference's own classes."""
    from multimodal_learning_amd.distiller_zoo import Similarity, feats_KL
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "zoo_sp_featskl.npz"))
    R = Report("distiller-zoo losses vs REFERENCE golden")
    for B in (8, 64):
        f_s = torch.as_tensor(g[f"f_s{B}"]).cuda().requires_grad_(True)
        f_t = torch.as_tensor(g[f"f_t{B}"]).cuda()
        l1 = Similarity()(f_s, f_t)
        g1, = torch.autograd.grad(l1.sum(), f_s)
        R.close(g[f"sp{B}"], l1, 1e-8, 1e-4, f"SP loss B={B}"); R.close(g[f"sp_g{B}"], g1, 1e-9, 2e-3, f"SP grad B={B}")
        l2 = feats_KL()(f_s, f_t)
        g2, = torch.autograd.grad(l2, f_s)
        R.close(np.asarray(g[f"fkl{B}"]).reshape(()), l2.reshape(()), 1e-6, 1e-4, f"feats_KL loss B={B}")
        R.close(g[f"fkl_g{B}"], g2, 1e-8, 1e-3, f"feats_KL grad B={B}")
    R.finish()


def test_distiller_zoo_rkd_and_pkt_vs_reference_golden(golden_dir):
    """Row f-4: RKDLoss and PKT (closed-form loss + gradient kernels, csrc/zoo.hip) against the reference's own classes
    at B = 8, 64 and 128; a loss scaled by a constant scales the gradient (the backward multiplies the stored gradient)."""
    from multimodal_learning_amd.distiller_zoo import RKDLoss, PKT
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "zoo_sp_featskl.npz"))
    R = Report("RKD / PKT vs REFERENCE golden")
    for B in (8, 64, 128):
        f_s = torch.as_tensor(g[f"r_f_s{B}"]).cuda().requires_grad_(True)
        f_t = torch.as_tensor(g[f"r_f_t{B}"]).cuda()
        l3 = RKDLoss()(f_s, f_t)
        g3, = torch.autograd.grad(3.0 * l3, f_s)
        R.close(np.asarray(g[f"rkd{B}"]).reshape(()), l3.reshape(()), 1e-7, 1e-4, f"RKD loss B={B}")
        R.close(3.0 * g[f"rkd_g{B}"], g3, 1e-8, 2e-3, f"RKD grad B={B}")
        l4 = PKT()(f_s, f_t)
        g4, = torch.autograd.grad(l4, f_s)
        R.close(np.asarray(g[f"pkt{B}"]).reshape(()), l4.reshape(()), 1e-9, 2e-3, f"PKT loss B={B}")
        R.close(g[f"pkt_g{B}"], g4, 1e-10, 5e-3, f"PKT grad B={B}")
    R.finish()


def test_superpixel_attention_masks_vs_reference_golden(golden_dir):
    """Row f-4: the tail of the MIA-2023 stage-1 `superpixel_attention_mask` (per-superpixel mean of the image gradient,
    top-Path_K superpixel mask, top-Omic_K omic mask) against the reference's own statements run on the same synthetic
    gradients (tests/golden/make_golden_superpixel.py).  Masks are integer work: exact."""
    import types
    import multimodal_learning_amd as m
    g = np.load(os.path.join(golden_dir, "superpixel_masks.npz"))
    for tag in ("a", "b"):
        grad = torch.as_tensor(g[f"{tag}_x_path_grad"]).cuda()
        og = torch.as_tensor(g[f"{tag}_x_omic_grad"]).cuda()
        sp = torch.as_tensor(g[f"{tag}_sp_mask"]).cuda()
        PK, OK = int(g[f"{tag}_Path_K"]), int(g[f"{tag}_Omic_K"])
        mask, mean = m.superpixel.superpixel_topk_mask(grad, sp, PK, return_mean=True)
        ref_mean = g[f"{tag}_mean"]
        assert mean.shape == ref_mean.shape
        assert np.abs(mean.cpu().numpy() - ref_mean).max() <= 2e-5 * np.abs(ref_mean).max()
        assert np.array_equal(mask.cpu().numpy(), g[f"{tag}_path_mask"]), tag
        assert np.array_equal(m.superpixel.omic_topk_mask(og, OK).cpu().numpy(), g[f"{tag}_omic_mask"]), tag
        opt = types.SimpleNamespace(Path_K=PK, Omic_K=OK)
        pm, om = m.superpixel.masks_from_input_gradients(opt, grad, og, sp, num_superpixels=ref_mean.shape[1])
        assert torch.equal(pm, mask) and om.shape == og.shape
        # bitwise reproducible (order-independent fixed-point accumulation)
        mask2, mean2 = m.superpixel.superpixel_topk_mask(grad, sp, PK, return_mean=True)
        assert torch.equal(mean, mean2) and torch.equal(mask, mask2)


def test_superpixel_attention_mask_end_to_end_vs_reference_function(golden_dir):
    """Row f-4 end to end: superpixel.superpixel_attention_mask (eval-mode PathomicNet forward, gradient of the fused NLL
    down to the image and the omic vector, aggregation, top-K masks) against the reference's own function run on the
    reference's network (tests/golden/make_golden_sp_attention.py).  Gradients at 1e-3 of their maximum (parity mode);
    the masks equal the reference's wherever the ranking is not decided by a margin below that tolerance."""
    import types
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    g = np.load(os.path.join(golden_dir, "sp_attention_b4_h64.npz"))
    B, H = int(g["B"]), int(g["H"])
    m.set_precision("bf16x6")
    try:
        opt = default_opt(dropout_rate=float(g["dropout_rate"]), cut_fuse_grad=False)
        model = m.define_net(opt, 1)
        model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
        model = model.cuda().train()
        bt = synthetic_batch(B, H, seed=620)
        sp = torch.as_tensor(g["sp_mask"])
        o = types.SimpleNamespace(Path_K=int(g["Path_K"]), Omic_K=int(g["Omic_K"]))
        # the gradients themselves
        model.eval()
        xp = bt["x_path"].cuda().requires_grad_(True); xo = bt["x_omic"].cuda().requires_grad_(True)
        pred = model(x_path=xp, x_grph=None, x_omic=xo)[5]
        cost = m.ops.NLLFn.apply(pred, bt["grade"].cuda(), float(B))
        gp, go = torch.autograd.grad(cost, [xp, xo])
        assert abs(float(cost) - float(g["cost"])) <= 1e-3 * abs(float(g["cost"]))
        for got, key in ((gp, "x_path_grad"), (go, "x_omic_grad")):
            ref = g[key]
            assert np.abs(got.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), key
        model.train()
        pm, om = m.superpixel.superpixel_attention_mask(o, None, model, bt["x_path"], torch.zeros(B), bt["x_omic"], sp,
                                                        bt["grade"], "cuda")
        assert model.training
        # masks: exact unless a rank is decided within the gradient tolerance
        _, mean_ref = m.superpixel.superpixel_topk_mask(torch.as_tensor(g["x_path_grad"]).cuda(), sp.cuda(), o.Path_K, return_mean=True)
        for b in range(B):
            if not np.array_equal(pm[b].cpu().numpy(), g["path_mask"][b]):
                srt = np.sort(mean_ref[b].cpu().numpy())[::-1]
                margin = srt[o.Path_K - 1] - srt[o.Path_K]
                assert margin <= 2e-3 * np.abs(srt).max(), (b, margin)
        assert (pm.cpu().numpy() != g["path_mask"]).mean() < 0.05
        assert (om.cpu().numpy() != g["omic_mask"]).mean() < 0.01
    finally:
        m.set_precision("bf16")


def test_cox_loss_vs_reference_golden(golden_dir):
    """utils.CoxLoss (one kernel: risk sets, log-sum, gradient) against the reference's own function and its autograd
    gradient (tests/golden/make_golden_cox.py), tied survival times included."""
    import multimodal_learning_amd as m
    g = np.load(os.path.join(golden_dir, "cox_loss.npz"))
    for B in (8, 64, 300):
        theta = torch.as_tensor(g[f"theta{B}"]).cuda().requires_grad_(True)
        loss = m.utils.CoxLoss(torch.as_tensor(g[f"t{B}"]), torch.as_tensor(g[f"c{B}"]), theta, "cuda")
        gr, = torch.autograd.grad(2.0 * loss, theta)
        assert abs(float(loss) - float(g[f"loss{B}"])) <= 1e-5 * abs(float(g[f"loss{B}"]))
        ref = 2.0 * g[f"g{B}"]
        assert gr.shape == ref.shape and np.abs(gr.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-8


def test_tsvd_update_aux_degenerate_inputs():
    """The proximal operator on degenerate stacks (both solvers): an all-zero stack stays zero, identical views collapse
    onto slice 0 (the other frequency slices vanish), a threshold above every singular value returns zero - no NaNs."""
    import multimodal_learning_amd as m
    from oracle import variants as OV
    for B in (32, 128):
        z = torch.zeros(B, B, 4)
        aux, tnn = m.tsvd.update_aux(z.cuda(), 0.1)
        assert torch.equal(aux.cpu(), z) and float(tnn) == 0.0
        g = torch.Generator().manual_seed(B)
        a = torch.rand(B, B, generator=g)
        same = a.unsqueeze(2).expand(B, B, 4).contiguous()
        aux, tnn = m.tsvd.update_aux(same.cuda(), 0.05)
        ref, tnn_ref = OV.update_aux(same, 0.05)
        assert torch.isfinite(aux).all()
        assert np.abs(aux.cpu().numpy() - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)
        assert abs(float(tnn) - tnn_ref) <= 1e-4 * max(abs(tnn_ref), 1.0)
        aux, tnn = m.tsvd.update_aux(same.cuda(), 1e4)
        assert float(aux.abs().max()) == 0.0 and float(tnn) == 0.0


def test_relational_losses_and_masks_on_degenerate_batches():
    """No NaNs where the reference's formulas have removable singularities: RKD / PKT with duplicated rows (zero distances,
    zero difference vectors), PKT with an all-zero row, superpixel masks with K = N and with unused labels."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.distiller_zoo import RKDLoss, PKT
    g = torch.Generator().manual_seed(1)
    f_t = torch.randn(16, 128, generator=g).relu_().cuda()
    f_s = torch.randn(16, 128, generator=g).relu_()
    f_s[5] = f_s[2]; f_s[9] = f_s[2]                      # duplicated rows
    f_s = f_s.cuda().requires_grad_(True)
    for crit in (RKDLoss(), PKT()):
        loss = crit(f_s, f_t)
        gr, = torch.autograd.grad(loss, f_s)
        assert torch.isfinite(loss) and torch.isfinite(gr).all()
    z = f_s.detach().clone(); z[3] = 0
    z.requires_grad_(True)
    loss = PKT()(z, f_t)
    gr, = torch.autograd.grad(loss, z)
    assert torch.isfinite(loss) and torch.isfinite(gr).all()
    grad = torch.randn(2, 3, 16, 16, generator=g).cuda()
    sp = (torch.arange(256).reshape(16, 16) // 64).unsqueeze(0).expand(2, 16, 16).contiguous()      # labels 0..3
    mask = m.superpixel.superpixel_topk_mask(grad, sp, 4, num_superpixels=4)                      # K = N: everything
    assert float(mask.min()) == 1.0
    mask, mean = m.superpixel.superpixel_topk_mask(grad, sp * 2, 2, num_superpixels=8, return_mean=True)   # odd labels unused
    assert torch.isfinite(mean).all() and float(mean[:, 1::2].abs().max()) == 0.0 and set(mask.unique().tolist()) <= {0.0, 1.0}


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["stage1", "v3"])
def test_vanilla_crd_criteria_draw_their_own_negatives(which):
    """`contrast_idx=None` in the stage-1 criterion (CRD_criterion.py:37-39) and in CRD_criterion_v3 (:37-39): the K + 1 bank rows
    per sample come from the AliasMethod table over uniform unigrams - the uniform draw (ph_alias_uniform_draw), column 0 := idx
    (VERDICT r05 missing 3).  The reference's stream is torch's CUDA generator, so the check is distributional + exact
    self-consistency: the loss and gradients equal those of a twin criterion fed the drawn rows explicitly; draws lie in the
    bank, differ from call to call and are uniform."""
    import copy
    from types import SimpleNamespace
    import multimodal_learning_amd as m
    from multimodal_learning_amd.CL_utils import memory_new as MN
    n_data, K, B = 4096, 512, 16
    torch.manual_seed(11)
    if which == "stage1":
        from multimodal_learning_amd.CL_utils import CRD_criterion as C
        crd = C.CRDLoss(SimpleNamespace(s_dim=128, t_dim=128, feat_dim=128, nce_k=K, nce_t=0.07, nce_m=0.5, n_data=n_data)).cuda()
        call = lambda c, fs, ft, idx, ci: c(fs, ft, idx, ci)            # noqa: E731
    else:
        from multimodal_learning_amd.CL_utils import CRD_criterion_v3 as C
        crd = C.CRDLoss(m.stage2_opt(nce_k=K), n_data).cuda()
        call = lambda c, fs, ft, idx, ci: c(0.7, fs, ft, idx, ci)       # noqa: E731
    crd.contrast.verbose = False
    twin = copy.deepcopy(crd)
    drawn = []
    orig = MN.draw_uniform_indices

    def spy(mem, y, width):
        out = orig(mem, y, width)
        drawn.append(out.clone())
        return out
    C.draw_uniform_indices = spy
    try:
        g = torch.Generator().manual_seed(5)
        for it in range(3):
            f_s = torch.randn(B, 128, generator=g).cuda().requires_grad_(True)
            f_t = torch.randn(B, 128, generator=g).cuda()
            idx = torch.randperm(n_data, generator=g)[:B].cuda()
            loss = call(crd, f_s, f_t, idx, None)
            ci = drawn[-1]
            assert tuple(ci.shape) == (B, K + 1) and ci.dtype == torch.int64
            assert torch.equal(ci[:, 0], idx) and int(ci.min()) >= 0 and int(ci.max()) < n_data
            f_s2 = f_s.detach().clone().requires_grad_(True)
            loss2 = call(twin, f_s2, f_t, idx, ci)
            assert torch.equal(loss, loss2) and torch.isfinite(loss).all()
            g1, = torch.autograd.grad(loss.sum(), f_s); g2, = torch.autograd.grad(loss2.sum(), f_s2)
            assert torch.equal(g1, g2) and float(g1.abs().max()) > 0
            assert torch.equal(crd.contrast.memory_v1, twin.contrast.memory_v1)
    finally:
        C.draw_uniform_indices = orig
    assert not torch.equal(drawn[0][:, 1:], drawn[1][:, 1:]) and not torch.equal(drawn[1][:, 1:], drawn[2][:, 1:])
    allv = torch.cat([d[:, 1:].reshape(-1) for d in drawn]).double()
    n = allv.numel()                                                   # 3 x 16 x 512 = 24 576 draws over 4096 rows
    assert abs(float(allv.mean()) - (n_data - 1) / 2) < 5 * n_data / (12 ** 0.5) / n ** 0.5
    counts = torch.bincount(allv.long(), minlength=n_data).double()
    chi2 = float(((counts - n / n_data) ** 2 / (n / n_data)).sum())
    assert abs(chi2 - (n_data - 1)) < 6 * (2 * (n_data - 1)) ** 0.5      # chi-square with n_data - 1 degrees of freedom
    assert int(crd.contrast._draw_step) == 3
