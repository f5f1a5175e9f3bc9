"""Data parallelism for the distill step: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference uses single-process nn.DataParallel which
re-broadcasts three networks every step (utils.py:257-260); here parameters stay resident and per step
there are exactly four exchanges:
  1. all-reduce (sum) of the flat gradient buffer, in buckets so RCCL pipelines over all xGMI links; the slice that is
     final after the backward of layers 4 and 3 (93 % of the bytes) starts there and overlaps the rest of the backward
  2. all-gather of (index, embed_s(f_s), embed_t(f_t)) rows -> identical CRD bank updates on every replica
  3. all-reduce of the 5x5 GK-Refine Gram matrix (global-batch gradient cosine)
  4. one-off all-reduce of the CRD normalisation sums (first batch only)
BatchNorm keeps per-replica statistics - that IS the DataParallel semantics of the reference.
Loss normalisers use the GLOBAL batch so summed gradients equal the single-process result."""
import torch
import torch.distributed as dist


class ReplicaSync:
    def __init__(self, group=None, bucket_bytes=32 << 20, grad_exchange="all_reduce"):
        """grad_exchange: how the flat gradient buffer is summed over the replicas -
          "all_reduce"      bucketed all-reduce (default);
          "reduce_scatter"  every bucket as reduce-scatter (each rank receives the sum of its 1/W shard) + all-gather of the
                            shards: the two halves of a ring all-reduce as separate collectives, so that on a point-to-point
                            xGMI fabric the first 8-GPU run can A/B the two schedules (`bench.py --grad-exchange`).  Same sums
                            up to the reduction order inside the library; a bucket's tail that does not divide by the world
                            size (< W elements) goes through a small all-reduce."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if grad_exchange not in ("all_reduce", "reduce_scatter"):
            raise ValueError("grad_exchange must be 'all_reduce' or 'reduce_scatter', got %r" % (grad_exchange,))
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.grad_exchange = grad_exchange
        self._shards = {}
        # gloo runs asynchronous collectives on a thread pool: a dependent pair must be chained on the host there; NCCL / RCCL
        # executes a group's collectives in issue order on its own stream
        self._ordered = dist.get_backend(group) != "gloo"

    # 1 -------------------------------------------------------------------------------------------------
    def _buckets(self, g, lo, hi):
        if self.grad_exchange == "all_reduce" or self.world_size == 1:
            return [dist.all_reduce(g[s:min(hi, s + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group,
                                    async_op=True) for s in range(lo, hi, self.bucket_elems)]
        W = self.world_size
        step = max(W, self.bucket_elems // W * W)
        works = []
        for s in range(lo, hi, step):
            e = min(hi, s + step)
            main = (e - s) // W * W
            if main:
                shard = self._shards.get((s, main))
                if shard is None:      # (one persistent shard buffer per bucket: a captured graph keeps its address)
                    shard = self._shards[(s, main)] = torch.empty(main // W, device=g.device, dtype=g.dtype)
                w1 = dist.reduce_scatter_tensor(shard, g[s:s + main], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if not self._ordered:
                    w1.wait()
                works.append(dist.all_gather_into_tensor(g[s:s + main], shard, group=self.group, async_op=True))
            if main < e - s:
                works.append(dist.all_reduce(g[s + main:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    def begin_grad_slice(self, flat, lo):
        """Start the all-reduce of flat.grad[lo:] and return at once.  Called from the trunk backward when the
        gradients of layers 3-4 and of everything behind them in the flat buffer (heads, CRD embeddings) are final
        (resnets._TrunkFn.backward, ph_resnet_backward_part): 93 % of the gradient bytes travel over xGMI while the
        backward of layers 2, 1 and the stem still runs.  `all_reduce_grads` then reduces the rest and waits for all."""
        g = flat if torch.is_tensor(flat) else flat.grad
        self._pending = (lo, self._buckets(g, lo, g.numel()))

    def all_reduce_grads(self, flat):
        g = flat if torch.is_tensor(flat) else flat.grad     # a FlatParams or a plain flat tensor
        hi, works = getattr(self, "_pending", None) or (g.numel(), [])
        self._pending = None
        works = works + self._buckets(g, 0, hi)
        for w in works:
            w.wait()
        return g

    # 2 -------------------------------------------------------------------------------------------------
    def all_gather_rows(self, y, v1, v2):
        W = self.world_size
        ys = [torch.empty_like(y) for _ in range(W)]
        dist.all_gather(ys, y.contiguous(), group=self.group)
        packed = torch.cat([v1, v2], dim=1).contiguous()
        ps = [torch.empty_like(packed) for _ in range(W)]
        dist.all_gather(ps, packed, group=self.group)
        allp = torch.cat(ps, dim=0)
        D = v1.shape[1]
        return torch.cat(ys, dim=0), allp[:, :D].contiguous(), allp[:, D:].contiguous()

    # 3 -------------------------------------------------------------------------------------------------
    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    # 4 -------------------------------------------------------------------------------------------------
    def all_reduce_z(self, sums, count):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        return count * self.world_size

    def all_gather_cat(self, t):
        """Rows of every replica, rank-ordered (the feature views of the t-SVD adjacency, train_test_tSVD.py:57-70)."""
        parts = [torch.empty_like(t) for _ in range(self.world_size)]
        dist.all_gather(parts, t.contiguous(), group=self.group)
        return torch.cat(parts, dim=0)

    def attach(self, step):
        """Wire the CRD memories of a DistillStep to this group and make every replica start identical."""
        self.attach_parts((step.criterion_kd, step.criterion_kd_path), (step.optimizer.flat.flat, step.ema_flat.flat),
                          (step.fix_model, step.model, step.ema_model))

    def attach_parts(self, crds, flats, modules):
        for crd in crds:
            crd.contrast.sync = self
            for b in (crd.contrast.memory_v1, crd.contrast.memory_v2, crd.contrast.params):
                dist.broadcast(b, src=0, group=self.group)
        for t in flats:
            dist.broadcast(t, src=0, group=self.group)
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)


def shard_batch(batch, rank, world_size):
    """Split a loader batch tuple (data_loaders_MT.py:256 layout) into this rank's contiguous shard."""
    (x_path, ema_x_path), x_grph, x_omic, censor, survtime, grade, index, sample_idx = batch
    B = x_path.shape[0]
    if B % world_size:
        raise ValueError("global batch must divide by the world size")
    n = B // world_size
    s = slice(rank * n, (rank + 1) * n)
    f = lambda t: t[s] if torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == B else t
    return ((f(x_path), f(ema_x_path)), f(x_grph), f(x_omic), f(censor), f(survtime), f(grade), f(index),
            f(sample_idx))


def comm_report(sync, device=None, backend=None):
    """What the communicator itself observed (bench.py's `comm` object; every rank must call it): the backend and its
    library version, the number of ranks that joined an all-reduce of ones, the PCI bus id of every rank's device (N
    distinct ones = one process per GPU) and the bucket plan of the gradient all-reduce.  Works under gloo on CPU
    (tests/test_dist_gloo.py)."""
    import os
    backend = backend or dist.get_backend(sync.group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    one = torch.ones(1, device=dev if dev.type == "cuda" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=sync.group)
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        bus = getattr(pr, "pci_bus_id", None)
        ident = ("%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), bus, getattr(pr, "pci_device_id", 0))) if bus is not None \
            else "%s#%d" % (pr.name, dev.index)
        ident = "%s (%s)" % (ident, getattr(pr, "uuid", ""))
    else:
        ident = "cpu pid %d" % os.getpid()
    ids = [None] * sync.world_size
    dist.all_gather_object(ids, ident, group=sync.group)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
    return {"backend": backend + (" (RCCL over xGMI)" if backend == "nccl" else ""), "library_version": ver,
            "world_size": sync.world_size, "ranks_joined_all_reduce": int(round(one.item())),
            "device_ids": ids, "distinct_devices": len(set(ids)),
            "grad_bucket_bytes": sync.bucket_elems * 4, "grad_exchange": sync.grad_exchange}
