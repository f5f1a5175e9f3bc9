#!/usr/bin/env python3
"""Golden vectors for the MIA-2022 stage-1 step with the t-SVD constraint (SURVEY row a16 end to end, BASELINE cfg 4
shape of the computation at a small size): the reference's own modules ("MIA 2022": networks_new, CL_utils.KD_losses)
driven in the order of train_test_tSVD.py:199-470 with `update_adj_tensor` compiled from that file.  The trainer's
`update_aux` lives in my_utils/TSVD_update_aux.py, which is NOT in the reference repository ("parity unpinned"): the
tensor-nuclear-norm proximal operator of oracle/variants.py stands in at exactly that call site (:382, :402).
Build container only.  Writes tests/golden/tsvd_step_b8_h64.npz (4 views) or, with the argument `8`,
tsvd_step_b8_h64_v8.npz (8 views: the mixed-feature views of :341-363)."""
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
REF = "/root/reference/MIA 2022"


def main(n_views=4, out_name="tsvd_step_b8_h64.npz"):
    from make_golden import install_shims, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.variants import update_aux
    sys.argv = ["x", "--model_name", "golden", "--reg_type", "none", "--beta1", "0.9", "--input_size_omic", "320",
                "--dropout_rate", "0", "--gpu_ids", "-1", "--checkpoints_dir", tempfile.mkdtemp(), "--batch_size", "8",
                "--tSVD_loss", "True", "--tSVD_mode", "pathomic", "--n_views", str(n_views), "--mu", "0.01", "--pho", "1.5",
                "--Lambda_global", "0.05", "--aux_iter", "1"]
    with contextlib.redirect_stdout(io.StringIO()):
        import options
        opt = options.parse_args()
        import networks_new as NN
        from CL_utils.KD_losses import pred_KD_loss
    opt.num_teachers = 2
    src = open(os.path.join(REF, "train_test_tSVD.py")).read()
    ns = {"torch": torch, "F": F}
    s0 = src.index("def update_adj_tensor"); s1 = src.index("def update_triplet_adj_tensor")
    exec(compile(src[s0:s1], "update_adj_tensor<reference>", "exec"), ns)
    update_adj_tensor = ns["update_adj_tensor"]
    with contextlib.redirect_stdout(io.StringIO()):
        model = NN.define_net(opt, 1); ema = NN.define_net(opt, 1)
    model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 4))        # a different mean teacher: non-trivial views
    for p in ema.parameters():
        p.detach_()
    optimizer = NN.define_optimizer(opt, model)
    model.train(); ema.train()
    B, nv = opt.batch_size, opt.n_views
    mu = opt.mu
    adj1 = [torch.zeros(B, B) for _ in range(nv)]; aux1 = [torch.zeros(B, B) for _ in range(nv)]
    adj2 = [torch.zeros(B, B) for _ in range(nv)]; aux2 = [torch.zeros(B, B) for _ in range(nv)]
    rec = dict(B=B, H=64, n_views=nv, mu=opt.mu, pho=opt.pho, max_mu=opt.max_mu, Lambda_global=opt.Lambda_global,
               lr=opt.lr, KD_weight=opt.KD_weight, cut_fuse_grad=int(bool(opt.cut_fuse_grad)))
    iter_num = 0
    for it in range(2):
        bt = synthetic_batch(B, 64, seed=70 + it)
        out = model(x_path=bt["x_path"], x_omic=bt["x_omic"])
        fuse_feat, path_feat, omic_feat, pred, pred_path, pred_omic = out[0], out[1], out[2], out[5], out[6], out[7]
        with torch.no_grad():
            eo = ema(x_path=bt["ema_x_path"], x_omic=bt["x_omic"])
        ema_fuse_feat, ema_path_feat, ema_omic_feat, ema_pred, ema_pred_path, ema_pred_omic = eo[0], eo[1], eo[2], eo[5], eo[6], eo[7]
        kd_fuse = pred_KD_loss(opt, pred, ema_pred)
        kd_path = (pred_KD_loss(opt, pred_path, ema_pred_path) + pred_KD_loss(opt, pred_path, ema_pred)) / 2.0
        kd_omic = (pred_KD_loss(opt, pred_omic, ema_pred_omic) + pred_KD_loss(opt, pred_omic, ema_pred)) / 2.0
        loss_kd = opt.KD_weight * (kd_fuse + kd_path + kd_omic)
        gr = bt["grade"]
        loss_nll = F.nll_loss(pred_path, gr) + F.nll_loss(pred_omic, gr) + F.nll_loss(pred, gr)
        loss = opt.lambda_nll * loss_nll + loss_kd
        # ---- :299-431
        feats1 = [fuse_feat.detach(), ema_fuse_feat, path_feat, ema_path_feat]
        feats2 = [fuse_feat.detach(), ema_fuse_feat, omic_feat, ema_omic_feat]
        if nv > 4:      # :305-307, :341-363: mixtures of the max-normalised mean-teacher features
            norm_path_feat = ema_path_feat / torch.max(ema_path_feat)
            norm_omic_feat = ema_omic_feat / torch.max(ema_omic_feat)
            for wa in (0.9, 0.8, 0.7, 0.6)[:nv - 4]:
                feats1.append(wa * norm_path_feat + (1 - wa) * norm_omic_feat)
                feats2.append(wa * norm_omic_feat + (1 - wa) * norm_path_feat)
        adj1 = update_adj_tensor(adj1, feats1)
        adj2 = update_adj_tensor(adj2, feats2)
        if it % opt.aux_iter == 0:
            for adj, aux_list, tag in ((adj1, aux1, "path"), (adj2, aux2, "omic")):
                stack = torch.stack([a.detach() for a in adj], dim=2)
                aux, tnn = update_aux(stack, opt.Lambda_global / mu)        # <- the absent my_utils module's call site
                aux = torch.as_tensor(aux).float()
                for v in range(nv):
                    aux_list[v] = aux[:, :, v]
                rec[f"{tag}_TNN{it}"] = tnn
            mu = min(mu * opt.pho, opt.max_mu)
        loss_tsvd = 0
        for v in range(nv):
            loss_tsvd += (mu / 2.0 * ((torch.norm(adj1[v] - aux1[v])) ** 2 + (torch.norm(adj2[v] - aux2[v])) ** 2))
        loss = loss + loss_tsvd
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        alpha = min(1 - 1 / (iter_num + 1), opt.ema_decay)
        for ep, p in zip(ema.parameters(), model.parameters()):
            ep.data.mul_(alpha).add_(p.data, alpha=1 - alpha)
        iter_num += 1
        rec.update({f"loss{it}": loss, f"loss_nll{it}": loss_nll, f"loss_kd{it}": loss_kd, f"loss_tsvd{it}": loss_tsvd,
                    f"mu{it}": mu, f"adj1_2_{it}": adj1[2], f"aux1_2_{it}": aux1[2], f"adj2_3_{it}": adj2[3],
                    f"aux2_0_{it}": aux2[0], f"pred{it}": pred, f"adj1_last_{it}": adj1[nv - 1],
                    f"aux2_last_{it}": aux2[nv - 1]})
        sd = model.state_dict()
        for k in ("omic_net.encoder.0.0.weight", "fusion.encoder2.0.weight", "path_net.fc_new1.0.weight"):
            rec[f"w{it}_{k}"] = sd[k].clone()
    np.savez_compressed(os.path.join(HERE, out_name), **npz(rec))
    print("wrote", out_name, [round(float(rec[f"loss{i}"]), 5) for i in range(2)],
          [round(float(rec[f"loss_tsvd{i}"]), 6) for i in range(2)])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "8":
        main(8, "tsvd_step_b8_h64_v8.npz")
    else:
        main()
