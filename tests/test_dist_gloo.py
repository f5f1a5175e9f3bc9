"""World-size-2 `gloo` tests (CPU) of the data-parallel exchanges of SURVEY.md section 8-e.  The kernels need a
GPU, so what is tested here is the host logic in multimodal_learning_amd.dist: sharding, bucketed gradient
all-reduce, the CRD row all-gather, the GK-Refine Gram all-reduce and the one-off Z reduction - each checked for
the algebraic identity that makes N replicas equal one big batch (sharded partials reduce to the global value
computed by the oracle's formulas)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0, "worker failed"
    return [ret[r] for r in range(world)]


def _grads(rank, world):
    from multimodal_learning_amd.dist import ReplicaSync
    sync = ReplicaSync(bucket_bytes=4096)          # forces several buckets
    g = torch.arange(5000, dtype=torch.float32) * (rank + 1)
    sync.all_reduce_grads(g)
    return g.sum().item(), g[1234].item()


def test_bucketed_grad_allreduce():
    out = _run(_grads)
    expect = torch.arange(5000, dtype=torch.float32) * 3
    for s, e in out:
        assert s == expect.sum().item() and e == expect[1234].item()


def _grads_reduce_scatter(rank, world):
    from multimodal_learning_amd.dist import ReplicaSync
    out = []
    for n, bucket_bytes in ((5000, 4096), (5001, 4096), (37, 1 << 20), (4096, 4096)):      # tails that do not divide by the world size
        ref = ReplicaSync(bucket_bytes=bucket_bytes)
        rs = ReplicaSync(bucket_bytes=bucket_bytes, grad_exchange="reduce_scatter")
        torch.manual_seed(100 * rank + n)
        g0 = torch.randn(n)
        a, b = g0.clone(), g0.clone()
        ref.all_reduce_grads(a)
        # two phases, as the step uses it: the tail slice starts while the head is not final
        b[:n // 3] = -7.0
        rs.begin_grad_slice(b, n // 3)
        b[:n // 3] = g0[:n // 3]
        rs.all_reduce_grads(b)
        out.append((a, b))
    return out


def test_reduce_scatter_gradient_exchange_equals_all_reduce():
    """`grad_exchange="reduce_scatter"` (reduce-scatter + all-gather per bucket, tails through a small
    all-reduce, two phases) gives the sums of the bucketed all-reduce on every rank - with two ranks a sum of two numbers,
    so bitwise."""
    res = _run(_grads_reduce_scatter)
    for case in range(4):
        a0, b0 = res[0][case]
        a1, b1 = res[1][case]
        assert torch.equal(a0, a1) and torch.equal(b0, b1), case      # identical on both ranks
        assert torch.equal(a0, b0), (case, (a0 - b0).abs().max().item())


def _grads_two_phase(rank, world):
    from multimodal_learning_amd.dist import ReplicaSync
    sync = ReplicaSync(bucket_bytes=4096)
    g = torch.arange(5000, dtype=torch.float32) * (rank + 1)
    g[:1700] = -1.0                                # "not final yet" when the tail slice starts its all-reduce
    sync.begin_grad_slice(g, 1700)                 # the trunk backward has passed layer 3
    g[:1700] = torch.arange(1700, dtype=torch.float32) * (rank + 1)   # ... layers 2, 1 and the stem finish
    sync.all_reduce_grads(g)
    again = g.clone()
    sync.all_reduce_grads(again)                   # no slice pending: the whole buffer is reduced
    return g.clone(), again


def test_grad_allreduce_in_two_phases_equals_one():
    out = _run(_grads_two_phase)
    expect = torch.arange(5000, dtype=torch.float32) * 3
    for g, again in out:
        assert torch.equal(g, expect) and torch.equal(again, expect * 2)


def _gather_cat(rank, world):
    from multimodal_learning_amd.dist import ReplicaSync
    sync = ReplicaSync()
    t = torch.full((3, 4), float(rank + 1))
    return sync.all_gather_cat(t)


def test_feature_view_allgather_is_rank_ordered():
    for out in _run(_gather_cat):
        assert out.shape == (6, 4) and torch.equal(out[:3], torch.ones(3, 4)) and torch.equal(out[3:], torch.full((3, 4), 2.0))


def _gather(rank, world):
    from multimodal_learning_amd.dist import ReplicaSync
    sync = ReplicaSync()
    B, D = 3, 128
    y = torch.arange(B) + 10 * rank
    v1 = torch.full((B, D), float(rank)); v2 = torch.full((B, D), float(rank) + 0.5)
    yy, a, b = sync.all_gather_rows(y, v1, v2)
    return yy.tolist(), a[:, 0].tolist(), b[:, 0].tolist(), a.shape, b.is_contiguous()


def test_crd_row_allgather_is_rank_ordered_and_identical():
    out = _run(_gather)
    assert out[0] == out[1]
    yy, a, b, shp, contig = out[0]
    assert yy == [0, 1, 2, 10, 11, 12] and a == [0, 0, 0, 1, 1, 1] and b == [0.5] * 3 + [1.5] * 3
    assert tuple(shp) == (6, 128) and contig


def _gram(rank, world):
    """GK-Refine under data parallelism: each rank holds d loss_i / d feat_s for ITS rows; the global-batch
    cosine Gram is the all-reduced sum of per-rank Grams (AEKD_loss :58-62 on the concatenated batch)."""
    from multimodal_learning_amd.dist import ReplicaSync
    sync = ReplicaSync()
    g = torch.Generator().manual_seed(0)
    G = torch.randn(5, 4 * 128, generator=g)            # whole batch of 4 samples
    local = G.view(5, 4, 128)[:, rank * 2:(rank + 1) * 2].reshape(5, -1)
    gram = (local @ local.T).reshape(-1).contiguous()
    sync.all_reduce_sum(gram)
    gram = gram.view(5, 5)
    nrm = gram.diag().sqrt()
    scale = (gram * 4 / (nrm[:, None] * nrm[None, :])).sum(1)
    ref_n = G.norm(dim=1, keepdim=True)
    ref = ((G @ G.T) * 4 / (ref_n @ ref_n.T)).sum(1)
    return torch.allclose(scale, ref, rtol=1e-5, atol=1e-6)


def test_gk_gram_allreduce_equals_global_batch():
    assert all(_run(_gram))


def _zsum(rank, world):
    from multimodal_learning_amd.dist import ReplicaSync
    sync = ReplicaSync()
    x = torch.arange(8, dtype=torch.float32).view(2, 4)[rank]         # this rank's scores
    sums = torch.stack([x.sum(), (2 * x).sum()])
    count = sync.all_reduce_z(sums, float(x.numel()))
    return (sums[0] / count).item(), (sums[1] / count).item()


def test_first_batch_z_uses_the_global_mean():
    for z1, z2 in _run(_zsum):
        assert z1 == pytest.approx(3.5) and z2 == pytest.approx(7.0)      # mean over BOTH ranks' scores


def test_shard_batch_layout():
    from multimodal_learning_amd.dist import shard_batch
    B = 8
    batch = ((torch.arange(B * 3).view(B, 3).float(), torch.zeros(B, 3)), torch.zeros(B), torch.ones(B, 5), torch.zeros(B),
             torch.zeros(B), torch.arange(B), torch.arange(B) + 100, torch.arange(B * 4).view(B, 4))
    s1 = shard_batch(batch, 1, 2)
    assert s1[0][0].shape == (4, 3) and s1[5].tolist() == [4, 5, 6, 7] and s1[6].tolist() == [104, 105, 106, 107]
    assert s1[7][0].tolist() == [16, 17, 18, 19]
    with pytest.raises(ValueError):
        shard_batch(batch, 0, 3)


def _loss_norm(rank, world):
    """Loss normalisers use the GLOBAL batch: per-rank partial losses divided by B_global and summed across ranks
    equal the single-process mean (DataParallel semantics of the reference's nll / KL / CRD losses)."""
    g = torch.Generator().manual_seed(1)
    pred = torch.log_softmax(torch.randn(6, 3, generator=g), 1)
    grade = torch.randint(0, 3, (6,), generator=g)
    ref = torch.nn.functional.nll_loss(pred, grade)
    lo = slice(rank * 3, rank * 3 + 3)
    part = -pred[lo][torch.arange(3), grade[lo]].sum() / 6.0           # what ph_nll_fwd computes with inv_bnorm = 1/6
    dist.all_reduce(part)
    return abs(part.item() - ref.item()) < 1e-6


def test_global_batch_loss_normaliser():
    assert all(_run(_loss_norm))


def _bench_cmd(*extra):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                           "--stub-step", *extra], env={**env, **_bench_cmd.env}, capture_output=True, text=True, timeout=240)


_bench_cmd.env = {}


def test_bench_launches_its_own_replicas_without_torchrun():
    """VERDICT r02 missing 1: `python bench.py --gpus N` (the driver's command line, no launcher, no WORLD_SIZE) must start
    the N replicas itself, reach init_process_group on every rank and print ONE JSON line from rank 0."""
    import json
    _bench_cmd.env = {}
    r = _bench_cmd()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["sum"] == 2 * 64.0 ** 3
    assert res["config"]["launch"] == "self-launched"
    # the `comm` object (dist.comm_report): what the communicator itself observed - the JSON line alone answers "did the
    # backend see N ranks, one device each" 
    c = res["comm"]
    assert c["backend"].startswith("gloo") and c["world_size"] == 2 and c["ranks_joined_all_reduce"] == 2
    assert len(c["device_ids"]) == 2 and c["distinct_devices"] == 2 and c["launch"] == "self-launched"
    assert c["grad_bucket_bytes"] == 32 << 20


def test_bench_launcher_falls_back_to_eager_when_the_graph_attempt_fails():
    import json
    _bench_cmd.env = {"PH_BENCH_STUB_FAIL_GRAPH": "1"}
    try:
        r = _bench_cmd()
    finally:
        _bench_cmd.env = {}
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "eager fallback" in json.loads(lines[0])["config"]["launch"]
    assert "ended with code 7" in r.stderr
