"""Two data-parallel replicas of the distillation step emulated on ONE GPU: two DistillStep
objects, one thread each, exchanging through `LocalSync` - an in-process stand-in with the interface of
`multimodal_learning_amd.dist.ReplicaSync` (bucketed gradient all-reduce, CRD row all-gather, Gram / Z reductions).
The real RCCL run is the driver's; what is pinned here is that the replicated step with global-batch normalisers
computes the same thing as one process on the whole batch, wherever that is defined:

  * BatchNorm keeps per-replica statistics (the DataParallel semantics of the reference), so the batch is built with
    the second shard's images and omic vectors equal to the first shard's: the forward statistics of a shard then equal
    those of the whole batch, while labels, bank indices and contrast indices differ per sample.
  * everything downstream of the student feature that couples samples only through normalisers / collectives must
    match the single process: the losses (summed over replicas), the GK-Refine weights, the all-reduced gradients of
    the grading head and of the four CRD embedding heads, and the updated bank rows.
  * the all-reduced flat gradient and the updated parameters are bitwise equal on the two replicas."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class LocalGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=60)
        self.slots = [None] * world


class LocalSync:
    def __init__(self, group, rank, expect_slice=True):
        self.g, self.rank, self.world_size = group, rank, group.world
        self.expect_slice = expect_slice      # DistillStep announces the layer 3-4 slice; the stage-1 step does not

    def _exchange(self, t):
        self.g.slots[self.rank] = t
        self.g.barrier.wait()
        parts = list(self.g.slots)
        self.g.barrier.wait()
        return parts

    def _sum(self, t):
        parts = self._exchange(t.clone())
        tot = parts[0].clone()
        for p in parts[1:]:
            tot += p                       # rank order on every replica: bitwise identical results
        t.copy_(tot)
        return t

    def begin_grad_slice(self, flat, lo):
        # Called in the middle of the trunk backward (layers 3-4 done).  The real ReplicaSync launches the asynchronous
        # all-reduce of the slice here; this stand-in snapshots the slice (what an all-reduce launched now would see) and
        # exchanges it later from the replica's own thread - both replicas' backward passes run on the ONE autograd
        # worker thread of the device, so blocking here would deadlock the emulation.
        g = flat if torch.is_tensor(flat) else flat.grad
        self.pending = (lo, g[lo:].clone())

    def all_reduce_grads(self, flat):
        g = flat if torch.is_tensor(flat) else flat.grad
        pend, self.pending = getattr(self, "pending", None), None
        if not self.expect_slice:
            assert pend is None
            return self._sum(g)
        assert pend is not None and 0 < pend[0] < g.numel(), "the trunk backward did not announce its finished slice"
        lo, snap = pend
        assert torch.equal(snap, g[lo:]), "gradients behind the announced offset changed after the announcement"
        self._sum(g[lo:])
        return self._sum(g[:lo])

    def all_reduce_sum(self, t):
        return self._sum(t)

    def all_reduce_z(self, sums, count):
        self._sum(sums)
        return count * self.world_size

    def all_gather_rows(self, y, v1, v2):
        ys, a, b = self._exchange(y.clone()), self._exchange(v1.clone()), self._exchange(v2.clone())
        return torch.cat(ys, 0), torch.cat(a, 0), torch.cat(b, 0)

    def all_gather_cat(self, t):
        return torch.cat(self._exchange(t.clone()), 0)

    def attach(self, step):
        for crd in (step.criterion_kd, step.criterion_kd_path):
            crd.contrast.sync = self

    def attach_parts(self, crds, flats, modules):
        for crd in crds:
            crd.contrast.sync = self


def _build(variant, sync, B, n_data, labels, **over):
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt
    kw = dict(batch_size=B, **over)
    if variant == "miccai2022":
        opt = default_opt(**kw)
        step = m.DistillStep(opt, n_data, device="cuda", sync=sync)
    elif variant == "mia2022":
        opt = default_opt(nce_k=512, grads_m=0.9, grads_thresh="False", thresh=0.1, **kw)
        step = m.DistillStep(opt, n_data, device="cuda", sync=sync, variant="mia2022")
    else:
        opt = default_opt(nce_k=512, nce_p=4, pos_extra="neighbors", neg_mode="all_others", start_reweight=0,
                          discrep_scale=1, max_discrep=2.0, use_grads_thresh="True", grads_thresh=0.0,
                          loss_weighting="GK_refine", **kw)
        class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
        step = m.DistillStep(opt, n_data, device="cuda", sync=sync, variant="mia2023", train_class_idx=class_idx)
    step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
    step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    g = torch.Generator().manual_seed(77)
    for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
        crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
        crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
        for bank in (crd.contrast.memory_v1, crd.contrast.memory_v2):
            bank.copy_((torch.rand(bank.shape, generator=g) * 2 - 1) * 0.15)
        crd.contrast.verbose = False
    return step, opt


def _batch(variant, B, H, n_data, labels):
    from oracle.step import synthetic_batch
    P, K = (300, 700) if variant == "miccai2022" else (1, 512)
    bt = synthetic_batch(B, H, n_data=n_data, P=P, K=K, seed=900)
    h = B // 2
    for k in ("x_path", "ema_x_path", "x_omic"):
        bt[k][h:] = bt[k][:h]              # per-shard BatchNorm statistics == whole-batch statistics
    if variant == "mia2023":
        bt["grade"] = labels[bt["index"]]
    z = torch.zeros(B)
    return ((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"])


def _head_grads(step):
    out = {}
    for name in ("fc_new2.weight", "fc_new2.bias"):
        out[name] = dict(step.model.named_parameters())[name].grad.detach().clone()
    for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
        for e in ("embed_s", "embed_t"):
            for pn, p in getattr(crd, e).named_parameters():
                out[f"crd{i}.{e}.{pn}"] = p.grad.detach().clone()
    return out


@pytest.mark.parametrize("variant", ["miccai2022", "mia2022", "mia2023"])
def test_two_replicas_equal_one_process_on_the_global_batch(variant):
    import multimodal_learning_amd as m
    from multimodal_learning_amd.dist import shard_batch
    B, H, n_data = 8, 64, 1024
    labels = torch.randint(0, 3, (n_data,), generator=torch.Generator().manual_seed(5))
    m.set_precision("bf16x6")
    try:
        batch = _batch(variant, B, H, n_data, labels)
        single, _ = _build(variant, None, B, n_data, labels)
        # the host-RNG rank draw of memory_new.py:311 is shared by the whole batch: the same list on every replica
        ranks = [np.arange(30, 50), np.arange(45, 65)] if variant == "miccai2022" else None
        o1 = single.step(batch, epoch=3, ranks=ranks)
        torch.cuda.synchronize()
        g1 = _head_grads(single)
        idx = batch[6].cuda()
        banks1 = [crd.contrast.memory_v1[idx].clone() for crd in (single.criterion_kd, single.criterion_kd_path)] + \
                 [crd.contrast.memory_v2[idx].clone() for crd in (single.criterion_kd, single.criterion_kd_path)]

        group = LocalGroup(2)
        reps = [_build(variant, LocalSync(group, r), B // 2, n_data, labels)[0] for r in range(2)]
        outs, errs = [None, None], []

        def run(r):
            try:
                torch.cuda.set_device(0)
                outs[r] = reps[r].step(shard_batch(batch, r, 2), epoch=3, ranks=ranks)
                torch.cuda.synchronize()
            except BaseException as e:      # noqa: BLE001 - re-raised in the main thread
                errs.append(e)
                group.barrier.abort()
        ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(300)
        if errs:
            raise errs[0]
        # replicas agree bitwise
        f0, f1 = reps[0].optimizer.flat, reps[1].optimizer.flat
        assert torch.equal(f0.grad, f1.grad) and torch.equal(f0.flat, f1.flat)
        assert torch.equal(reps[0].ema_flat.flat, reps[1].ema_flat.flat)
        # ... and equal the single process where that is defined
        for k in ("loss", "loss_cls", "loss_div1", "loss_div2", "loss_kd1", "loss_kd2"):
            tot = sum(float(o[k]) for o in outs)
            ref = float(o1[k])
            assert abs(tot - ref) <= 2e-4 * max(abs(ref), 1e-2), (k, tot, ref)
        if o1.get("scale") is not None:
            s1, s2 = o1["scale"].cpu().numpy(), outs[0]["scale"].cpu().numpy()
            assert np.abs(s1 - s2).max() <= 2e-4 * np.abs(s1).max(), (s1, s2)
            assert torch.equal(outs[0]["scale"], outs[1]["scale"])
        g2 = _head_grads(reps[0])
        gmax = max(float(v.abs().max()) for v in g1.values())
        for k, v in g1.items():
            err = float((g2[k] - v).abs().max())
            assert err <= 1e-3 * float(v.abs().max()) + 1e-6 * gmax, (k, err, float(v.abs().max()))
        banks2 = [crd.contrast.memory_v1[idx].clone() for crd in (reps[0].criterion_kd, reps[0].criterion_kd_path)] + \
                 [crd.contrast.memory_v2[idx].clone() for crd in (reps[0].criterion_kd, reps[0].criterion_kd_path)]
        for a, b in zip(banks1, banks2):
            assert float((a - b).abs().max()) <= 1e-5
        for c0, c1 in zip((reps[0].criterion_kd, reps[0].criterion_kd_path), (reps[1].criterion_kd, reps[1].criterion_kd_path)):
            assert torch.equal(c0.contrast.memory_v1, c1.contrast.memory_v1)
            assert torch.equal(c0.contrast.params, c1.contrast.params)
    finally:
        m.set_precision("bf16")


def test_two_replicas_with_the_l1_regulariser_stay_identical_and_average_their_gradients():
    """with --reg_type all the trunk gradients reach the flat buffer through AccumulateGrad AFTER the point
    where the overlapped all-reduce of the layer-3/4 slice used to start, so that slice must not be announced early
    (LocalSync(expect_slice=False) asserts no announcement arrives), the replicas must end bitwise equal, and the head
    gradients must equal the single process on the global batch plus lambda_reg * sgn(W) once (not once per replica)."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.dist import shard_batch
    B, H, n_data = 8, 64, 1024
    labels = torch.randint(0, 3, (n_data,), generator=torch.Generator().manual_seed(5))
    m.set_precision("bf16x6")
    try:
        batch = _batch("miccai2022", B, H, n_data, labels)
        ranks = [np.arange(30, 50), np.arange(45, 65)]
        single, opt = _build("miccai2022", None, B, n_data, labels, reg_type="all")
        single.step(batch, epoch=3, ranks=ranks)
        torch.cuda.synchronize()
        g1 = _head_grads(single)
        group = LocalGroup(2)
        reps = [_build("miccai2022", LocalSync(group, r, expect_slice=False), B // 2, n_data, labels, reg_type="all")[0]
                for r in range(2)]
        assert all(getattr(r.model, "_grad_ready_hook", None) is None for r in reps)
        errs = []

        def run(r):
            try:
                torch.cuda.set_device(0)
                reps[r].step(shard_batch(batch, r, 2), epoch=3, ranks=ranks)
                torch.cuda.synchronize()
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
                group.barrier.abort()
        ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(300)
        if errs:
            raise errs[0]
        f0, f1 = reps[0].optimizer.flat, reps[1].optimizer.flat
        assert torch.equal(f0.grad, f1.grad) and torch.equal(f0.flat, f1.flat)
        g2 = _head_grads(reps[0])
        # the replicas SUM their gradients; the data terms add up to the single process' through the global-batch
        # normalisers and the L1 term (lambda_reg / world_size per replica) to lambda_reg * sgn(W) once - g1 holds it too
        for k in ("fc_new2.weight", "fc_new2.bias"):
            err = float((g2[k] - g1[k]).abs().max())
            assert err <= 1e-3 * float(g1[k].abs().max()) + 2e-5, (k, err)
        w = dict(single.model.named_parameters())["fc_new2.weight"].detach()
        assert float(g1["fc_new2.weight"].abs().max()) > 0 and opt.lambda_reg > 0 and float(w.abs().max()) > 0
    finally:
        m.set_precision("bf16")


def test_split_trunk_backward_is_bitwise_the_single_call():
    """ph_resnet_backward_part (layers 4-3, then layers 2-1 + stem, with a hook in between) writes bitwise the gradients
    of ph_resnet_backward; at the hook the gradients of layers 3-4 already hold their final values."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt
    opt = default_opt()
    x = (torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(3)) * 2 - 1).cuda()
    res = []
    for split in (False, True):
        net = m.define_net(opt, 1, path_only=True).cuda()
        net.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
        net.train()
        seen = {}
        if split:
            l4 = net.layer4[1].conv2.weight
            net._grad_ready_hook = lambda: seen.update(at_hook="called")
        out = net(x_path=x)
        (out[1].square().sum() + out[0].sum()).backward()
        torch.cuda.synchronize()
        if split:
            assert seen.get("at_hook") == "called"
        res.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    assert res[0].keys() == res[1].keys() and len(res[0]) > 60
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def test_two_stage1_replicas_with_tsvd_orth_crd_equal_one_process():
    """TeacherStage1Step under data parallelism (BASELINE configs 4 / 5 run their stage-1 trainers on several GPUs): the
    t-SVD adjacency / auxiliary tensors span the global batch (all-gathered feature views), the orthogonality loss sees
    the all-reduced cross-correlation, the vanilla CRD bank and the loss normalisers use the global batch."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd.dist import shard_batch
    from oracle import weights as W
    from oracle.step import synthetic_batch
    B, H, n_data, K = 8, 64, 1024, 256

    def build(bs, sync):
        opt = m.stage2_opt(dropout_rate=0.0, batch_size=bs, cut_fuse_grad=True, num_teachers=2)
        opt.pred_distill, opt.KD_weight, opt.SP_distill = 1, 1.0, 0
        opt.CRD_distill, opt.CRD_weight, opt.nce_k, opt.n_data = 1, 0.1, K, n_data
        opt.orth_loss = "True"
        opt.tSVD_loss, opt.tSVD_mode, opt.n_views, opt.aux_iter = "True", "pathomic", 4, 1
        opt.mu, opt.pho, opt.max_mu, opt.Lambda_global = 0.01, 1.5, 1.0, 0.05
        model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
        model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3)); ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 4))
        st = m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda()), sync=sync)
        g = torch.Generator().manual_seed(11)
        for crd in (st.CRD_criterion_path, st.CRD_criterion_omic, st.CRD_criterion_fuse):
            for i, e in enumerate((crd.embed_s, crd.embed_t)):
                for p in e.parameters():
                    p.data.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.1)
            for bank in (crd.contrast.memory_v1, crd.contrast.memory_v2):
                bank.copy_((torch.rand(bank.shape, generator=g) * 2 - 1) * 0.15)
            crd.contrast.verbose = False
        return st

    heads = ("classifier.0.weight", "path_net.fc_new2.weight", "omic_net.classifier.0.weight")
    m.set_precision("bf16x6")
    try:
        bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=950)
        for k in ("x_path", "ema_x_path", "x_omic"):
            bt[k][B // 2:] = bt[k][:B // 2]
        z = torch.zeros(B)
        batch = ((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"])
        single = build(B, None)
        names = dict(single.model.named_parameters())
        assert all(h in names for h in heads), [n for n in names if "classifier" in n or "fc_new2" in n]
        o1 = single.step(batch)
        g1 = {h: names[h].grad.clone() for h in heads}
        group = LocalGroup(2)
        reps = [build(B // 2, LocalSync(group, r, expect_slice=False)) for r in range(2)]
        outs, errs = [None, None], []

        def run(r):
            try:
                outs[r] = reps[r].step(shard_batch(batch, r, 2))
                torch.cuda.synchronize()
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
                group.barrier.abort()
        ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(300)
        if errs:
            raise errs[0]
        f0, f1 = reps[0].optimizer.flat, reps[1].optimizer.flat
        assert torch.equal(f0.grad, f1.grad) and torch.equal(f0.flat, f1.flat)
        for k in ("loss_tsvd", "loss_orth"):          # functions of the global batch: the same value everywhere
            assert torch.equal(outs[0][k], outs[1][k])
            assert abs(float(outs[0][k]) - float(o1[k])) <= 2e-4 * max(abs(float(o1[k])), 1e-3), (k, float(outs[0][k]), float(o1[k]))
        for k in ("loss_nll", "loss_pred_KD", "loss_CRD"):      # partial sums over the replica's rows
            tot = sum(float(o[k]) for o in outs)
            assert abs(tot - float(o1[k])) <= 2e-4 * max(abs(float(o1[k])), 1e-3), (k, tot, float(o1[k]))
        for v in range(4):
            assert float((reps[0].aux_tensor1[v] - single.aux_tensor1[v]).abs().max()) <= 2e-5
            assert float((reps[0].adj_tensor2[v] - single.adj_tensor2[v]).abs().max()) <= 2e-5
            assert torch.equal(reps[0].aux_tensor2[v], reps[1].aux_tensor2[v])
        n0 = dict(reps[0].model.named_parameters())
        for h in heads:
            err = float((n0[h].grad - g1[h]).abs().max())
            assert err <= 1e-3 * float(g1[h].abs().max()) + 1e-7, (h, err)
    finally:
        m.set_precision("bf16")


def test_replica_sync_step_is_captured_in_one_graph_with_rccl_inside():
    """with a ReplicaSync attached, the fused loss head must not wait on its side stream (it never
    forks it then) - a wait on a stream outside the capture invalidates the capture.  One rank, RCCL: `bench.py --force-dist`
    runs the data-parallel code path (bucketed gradient all-reduce in two phases, bank-row all-gathers, Gram all-reduce) INSIDE
    the captured step, at the per-GPU shape of BASELINE configs[2] (32 tiles of 512 x 512 = global batch 256 over 8 GPUs); the line
    must say the step was replayed from a graph and carry the communicator's own observations."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    # 32 tiles of 512 x 512 = the per-GPU leg of BASELINE configs[2] (global batch 256 over 8 GPUs) through the replica-sync path
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--batch", "32", "--size", "512", "--steps", "3",
                        "--no-parity-mode", "--no-variants", "--no-cpu-baseline", "--no-north-star-block"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["config"]["launch"].startswith("one captured HIP graph per step"), res["config"]["launch"]
    c = res["comm"]
    assert c["backend"].startswith("nccl") and c["ranks_joined_all_reduce"] == 1 and c["distinct_devices"] == 1
    assert c["launch"].startswith("collectives captured inside the step's HIP graph")
    assert c["grad_allreduce_bytes"] > 40e6 and 0.0 <= c["exposed_grad_allreduce_ms"] < 5.0


def test_stage1_tsvd_step_is_captured_with_rccl_inside():
    """The stage-1 trainer with the t-SVD term under a process group (one rank, RCCL): the feature all-gathers of the global
    adjacency tensors and the gradient all-reduce are captured inside TeacherStage1Step's step graph; the variant line of
    `bench.py --force-dist --variant tsvd` must come from graph replay (a failed capture would fall back to eager and say so)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--variant", "tsvd", "--batch", "16", "--size", "128",
                        "--steps", "4", "--warmup", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["config"]["launch"] == "one captured HIP graph per step", res["config"]
    assert np.isfinite(res["config"]["final_loss"])
