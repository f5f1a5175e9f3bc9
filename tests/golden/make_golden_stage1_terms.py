#!/usr/bin/env python3
"""Golden vectors for the optional terms of the stage-1 mean-teacher trainer, produced by running the
reference's own modules: MICCAI-2022/CL_utils/orthogonal_loss.OrthLoss, CL_utils/CRD_criterion.CRDLoss (vanilla bank,
two-layer heads) and the batch body of train_test_MT.py:121-230 with --CRD_distill 1 --orth_loss True for two steps.
Build container only.  Writes tests/golden/stage1_terms.npz and stage1_terms_step_b4_h64.npz."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MICCAI-2022"


def embed2_state(seed):
    """Seed recipe for the two-layer projection head (CRD_criterion.py:219-234): no weight blobs in the fixture."""
    g = torch.Generator().manual_seed(seed)
    return {"linear.0.weight": torch.randn(128, 128, generator=g) * 0.08, "linear.0.bias": torch.randn(128, generator=g) * 0.02,
            "linear.2.weight": torch.randn(128, 128, generator=g) * 0.08, "linear.2.bias": torch.randn(128, generator=g) * 0.02}


def main():
    from make_golden import install_shims, ref_opt, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.variants import CRDv3State
    opt = ref_opt(tempfile.mkdtemp(), extra=["--nce_k", "512", "--orth_loss", "True", "--CRD_distill", "1", "--n_data", "1024"])
    opt.cut_fuse_grad = False
    opt.num_teachers = 2
    with contextlib.redirect_stdout(io.StringIO()):
        import networks_new as NN
        from CL_utils.KD_losses import pred_KD_loss
        from CL_utils.CRD_criterion import CRDLoss
        from CL_utils.orthogonal_loss import OrthLoss
    n_data, K = opt.n_data, opt.nce_k

    # ---------------- (i) module-level vectors
    g = torch.Generator().manual_seed(9)
    rec = dict(K=K, n_data=n_data)
    x1 = torch.randn(8, 128, generator=g).relu_().requires_grad_(True)
    x2 = (torch.randn(8, 128, generator=g) * 0.7).requires_grad_(True)
    lo = OrthLoss()(x1, x2)
    g1, g2 = torch.autograd.grad(lo, [x1, x2])
    rec.update(orth_x1=x1, orth_x2=x2, orth_loss=lo, orth_g1=g1, orth_g2=g2)
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        crd = CRDLoss(opt)
    crd.embed_s.load_state_dict(embed2_state(70)); crd.embed_t.load_state_dict(embed2_state(71))
    st = CRDv3State(n_data, K=K, seed=80)
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    for it in range(2):
        f_s = torch.randn(8, 128, generator=g).relu_().requires_grad_(True)
        f_t = torch.randn(8, 128, generator=g).relu_()
        index = torch.randperm(n_data, generator=g)[:8]
        sidx = torch.randint(0, n_data, (8, K + 1), generator=g); sidx[:, 0] = index
        with contextlib.redirect_stdout(io.StringIO()):
            loss = crd(f_s, f_t, index, sidx)
        gs = torch.autograd.grad(loss.sum(), [f_s, crd.embed_s.linear[0].weight, crd.embed_s.linear[2].weight,
                                              crd.embed_t.linear[2].bias])
        rec.update({f"f_s{it}": f_s, f"f_t{it}": f_t, f"index{it}": index, f"sidx{it}": sidx, f"loss{it}": loss,
                    f"g_fs{it}": gs[0], f"g_w0{it}": gs[1], f"g_w2{it}": gs[2], f"g_tb2{it}": gs[3],
                    f"params{it}": crd.contrast.params.clone(), f"bank_v1_rows{it}": crd.contrast.memory_v1[index].clone(),
                    f"bank_v2_rows{it}": crd.contrast.memory_v2[index].clone()})
    np.savez_compressed(os.path.join(HERE, "stage1_terms.npz"), **npz(rec))

    # ---------------- (ii) two steps of the batch body with both terms on
    with contextlib.redirect_stdout(io.StringIO()):
        model = NN.define_net(opt, 1)
        ema = NN.define_net(opt, 1)
        crds = [CRDLoss(opt) for _ in range(3)]          # path, omic, fuse (train_test_MT.py:74-76)
    model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    for p in ema.parameters():
        p.detach_()
    for i, c in enumerate(crds):
        c.embed_s.load_state_dict(embed2_state(90 + 2 * i)); c.embed_t.load_state_dict(embed2_state(91 + 2 * i))
        sti = CRDv3State(n_data, K=K, seed=100 + i)
        c.contrast.memory_v1.copy_(sti.memory_v1); c.contrast.memory_v2.copy_(sti.memory_v2)
    ml = torch.nn.ModuleList([model])
    for c in crds:
        ml.append(c.embed_s); ml.append(c.embed_t)       # :84-90
    optimizer = NN.define_optimizer(opt, ml)
    orth = OrthLoss()
    ml.train(); ema.train()

    def update_ema_variables(model, ema_model, alpha, global_step):     # train_test_MT.py:34-38
        alpha = min(1 - 1 / (global_step + 1), alpha)
        for ema_param, param in zip(ema_model.parameters(), model.parameters()):
            ema_param.data.mul_(alpha).add_(param.data, alpha=1 - alpha)

    rec = dict(B=4, H=64, K=K, n_data=n_data, CRD_weight=opt.CRD_weight, KD_weight=opt.KD_weight)
    iter_num = 0
    for it in range(2):
        bt = synthetic_batch(4, 64, n_data=n_data, P=1, K=K, seed=60 + it)
        out = model(x_path=bt["x_path"], x_omic=bt["x_omic"])
        fuse_feat, path_feat, omic_feat, pred, pred_path, pred_omic = out[0], out[1], out[2], out[5], out[6], out[7]
        with torch.no_grad():
            eo = ema(x_path=bt["ema_x_path"], x_omic=bt["x_omic"])
        ema_fuse_feat, ema_pred, ema_pred_path, ema_pred_omic = eo[0], eo[5], eo[6], eo[7]
        with contextlib.redirect_stdout(io.StringIO()):
            loss_CRD = opt.CRD_weight * crds[2](fuse_feat, ema_fuse_feat.detach(), bt["index"], bt["sample_idx"])
        kd_fuse = pred_KD_loss(opt, pred, ema_pred)
        kd_path = (pred_KD_loss(opt, pred_path, ema_pred_path) + pred_KD_loss(opt, pred_path, ema_pred)) / 2.0
        kd_omic = (pred_KD_loss(opt, pred_omic, ema_pred_omic) + pred_KD_loss(opt, pred_omic, ema_pred)) / 2.0
        loss_kd = opt.KD_weight * (kd_fuse + kd_path + kd_omic)
        gr = bt["grade"]
        loss_nll = F.nll_loss(pred_path, gr) + F.nll_loss(pred_omic, gr) + F.nll_loss(pred, gr)
        loss = opt.lambda_nll * loss_nll + loss_CRD + loss_kd
        loss_orth = orth(path_feat, omic_feat)
        loss = loss + loss_orth
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        update_ema_variables(model, ema, opt.ema_decay, iter_num)
        iter_num += 1
        rec.update({f"loss{it}": loss, f"loss_nll{it}": loss_nll, f"loss_kd{it}": loss_kd, f"loss_CRD{it}": loss_CRD,
                    f"loss_orth{it}": loss_orth, f"pred{it}": pred,
                    f"bank_v1_rows{it}": crds[2].contrast.memory_v1[bt["index"]].clone()})
        sd, esd = model.state_dict(), ema.state_dict()
        for k in ("omic_net.encoder.0.0.weight", "fusion.encoder2.0.weight", "classifier.0.weight",
                  "path_net.fc_new1.0.weight"):
            rec[f"w{it}_{k}"] = sd[k].clone(); rec[f"e{it}_{k}"] = esd[k].clone()
        rec[f"w{it}_embed_s_fuse"] = crds[2].embed_s.linear[2].weight.detach().clone()
        rec[f"w{it}_embed_s_path"] = crds[0].embed_s.linear[0].weight.detach().clone()    # unused head: weight decay only
    np.savez_compressed(os.path.join(HERE, "stage1_terms_step_b4_h64.npz"), **npz(rec))
    print("wrote stage1_terms*.npz", [round(float(rec[f"loss{i}"]), 5) for i in range(2)],
          [round(float(rec[f"loss_CRD{i}"]), 5) for i in range(2)], [round(float(rec[f"loss_orth{i}"]), 6) for i in range(2)])


if __name__ == "__main__":
    main()
