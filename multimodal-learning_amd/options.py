"""Option namespace shim.  The reference's flag system (MICCAI-2022/options.py:8-164) is harness code that can
drive these modules unchanged (they read the same `opt.*` attribute names, string booleans included); this
helper only builds a Namespace with the reference DEFAULTS overlaid with the README stage-2 command
(MICCAI-2022/README.md:30-33) for callers that do not go through argparse (bench.py, tests, smoke)."""
from types import SimpleNamespace


def stage2_opt(**overrides):
    o = SimpleNamespace(
        # model (options.py:108-151)
        mode="pathomic", task="grad", act_type="LSM", init_type="max", init_gain=0.02, gpu_ids=[],
        path_dim=128, omic_dim=128, mmhid=128, label_dim=3, input_size_omic=320, input_size_path=512,
        dropout_rate=0.1, fusion_type="pofusion", skip=0, use_bilinear=1, path_gate=1, omic_gate=1,
        path_scale=1, omic_scale=1, return_grad="False", cut_fuse_grad=True,
        # distillation (options.py:27-90) at the README stage-2 values
        distill="crd", alpha=1.0, beta=0.02, kd_T=1.0, num_teachers=2, which_teacher="fuse",
        CE_grads=True, assign_weights="True", reg_type="none", sample_KD="False",
        s_dim=128, t_dim=128, feat_dim=128, nce_p=300, nce_p2=20, nce_k=700, nce_k2=512, nce_t=0.07, nce_m=0.5,
        select_pos_pairs=True, select_neg_pairs="True", select_pos_mode="mid", n_data=1024,
        # optimiser / schedule (options.py:125-137,155-158)
        optimizer_type="adam", lr=5e-4, beta1=0.9, beta2=0.999, weight_decay=4e-4, lr_policy="linear", niter=0,
        niter_decay=30, epoch_count=1, lambda_cox=1.0, lambda_nll=1.0, lambda_reg=3e-4, ema_decay=0.99,
        global_step=0, batch_size=16,
        grads_m=0.9, grads_thresh="False", thresh=0.0,   # MIA-2022 momentum GK-Refine ("MIA 2022/options.py":80-82)
        pos_extra="neighbors", use_grads_thresh="True", max_discrep=1, discrep_scale=1, start_reweight=0,   # MIA-2023
        overlap_teachers=True)     # ours: run the EMA / teacher forwards on a second HIP stream
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError("unknown option %r" % k)
        setattr(o, k, v)
    return o
