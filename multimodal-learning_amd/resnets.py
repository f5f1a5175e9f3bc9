"""Drop-in for the reference's MICCAI-2022/resnets.py (ResNet / BasicBlock / ResNet18): same module tree,
same ``state_dict`` keys (OIHW fp32 at the boundary), same ``forward(**kwargs)`` 5-tuple
(resnets.py:267-272) - but the trunk forward AND backward run as hand-written HIP kernels through
``ph_resnet_forward`` / ``ph_resnet_backward`` (include/pathomic_hip.h).

The torch sub-modules (nn.Conv2d, nn.BatchNorm2d, ...) are used ONLY as parameter containers so that
checkpoints round-trip; their ``forward`` is never called.
"""
import ctypes as C

import torch
import torch.nn as nn
from torch.nn import Parameter

from . import ops
from ._lib import lib, check, ptr, stream, require_cuda

__all__ = ["ResNet", "ResNet18", "BasicBlock"]


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    """Parameter container with the reference layout (resnets.py:37-56)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):   # pragma: no cover
        raise RuntimeError("BasicBlock is executed by the fused HIP trunk, not standalone")


class _Plan:
    """RAII wrapper of a PhResnetPlan."""

    def __init__(self, B, H, W, prec):
        self.h = lib().ph_resnet_plan_create(B, H, W, prec)
        if not self.h:
            raise RuntimeError(f"ph_resnet_plan_create({B},{H},{W},{prec}) failed")
        self.ws_bytes = lib().ph_resnet_workspace_bytes(self.h)
        self.packed_bytes = lib().ph_resnet_packed_bytes(self.h)
        self.key = (B, H, W, prec)

    def __del__(self):
        try:
            if self.h:
                lib().ph_resnet_plan_destroy(self.h)
        except Exception:
            pass


class _TrunkFn(torch.autograd.Function):
    """x [B,3,H,W] f32 -> (f3 [B,256], f4 [B,512]); backward fills every trunk parameter's gradient."""

    @staticmethod
    def forward(ctx, x, net, *params):
        x = require_cuda(x, "x_path")
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        B, Cc, H, W = x.shape
        if Cc != 3:
            raise RuntimeError("x_path must be [B,3,H,W]")
        # (`needs_input_grad` mirrors requires_grad even under torch.no_grad(), and grad mode is always off inside
        # Function.forward: the caller's grad mode is recorded by _forward_impl - a forward under no_grad is forward-only)
        fwd_only = not (getattr(net, "_caller_grad_mode", True) and any(ctx.needs_input_grad))
        want_dx = ctx.needs_input_grad[0]
        prec = ops.get_precision(fwd_only)
        if prec == ops.PREC_FP16X3 and (want_dx or ((H | W) & 3)):
            # the half-pair arithmetic is built for image sizes that are multiples of 4 (the pooled form of the stem's
            # BatchNorm-backward sums) and has no image gradient: such forwards (odd evaluation sizes; the MIA-2023 superpixel
            # attention, train_test_MT_SP_Masking.py:62-75) run in the fp32-equivalent split-plane arithmetic instead, which
            # meets the same tolerance (they used to fail inside ph_resnet_plan_create / ph_resnet_backward_input)
            if net.training and any(ctx.needs_input_grad) and not fwd_only:
                raise NotImplementedError("precision 'fp16x3': a TRAINING forward needs H and W to be multiples of 4 (got %dx%d); "
                                          "use set_precision('bf16x6') for this size" % (H, W))
            prec = ops.PREC_BF16X6
        plan = net._get_plan(B, H, W, prec)
        packed = net._get_packed(plan)
        table = net._param_table()
        if want_dx and net.training:
            raise NotImplementedError("gradient with respect to the image in train mode (the reference only takes it "
                                      "through an eval-mode net: train_test_MT_SP_Masking.py:62-75)")
        # One workspace per network serves forward and backward (grad mode is off inside Function.forward, so this was
        # always the cached one).  A trainer that runs the SAME network several times before one backward (the MIA-2023
        # masked views, train_test_MT_SP_Masking.py:204-208) sets `net._multi_forward`: every taped forward then keeps a
        # workspace of its own.
        need_grad = any(ctx.needs_input_grad) and getattr(net, "_multi_forward", False)
        ws = net._get_workspace(plan, persistent=need_grad)
        f3 = torch.empty(B, 256, device=x.device, dtype=torch.float32)
        f4 = torch.empty(B, 512, device=x.device, dtype=torch.float32)
        # flags of ph_resnet_forward: train / eval; +4 when no backward will follow (no gradient is recorded: the EMA and
        # teacher networks of the distillation step) - the trunk then fuses bn1 + ReLU into conv2's operand staging;
        # +8 (`net._no_fuse`, an A/B and test switch) keeps the separate passes; +16 (`net._no_masked`, likewise) keeps the
        # first-generation kernel for the 3x3 stride-2 convolutions
        # +32 (`net._no_stem_pool`, likewise) keeps the separate stem conv and pooling passes in forward-only networks
        flags = (1 if net.training else 2) | (4 if fwd_only else 0) | (8 if getattr(net, "_no_fuse", False) else 0) | \
            (16 if getattr(net, "_no_masked", False) else 0) | (32 if getattr(net, "_no_stem_pool", False) else 0)
        # an image that has been packed already (pack_shared_input: the student and the teacher of the distillation step read
        # the same x_path) enters as its NHWC4 tensor, flag +64; the tensor is kept alive for the backward's stem wgrad
        shared = getattr(net, "_x4_shared", None)
        x4 = None
        if shared is not None:
            net._x4_shared = None      # single use: a later forward on recycled memory at the same address must not find it
            if shared[0] == (x.data_ptr(), tuple(x.shape), _x4_format(plan.key[3])):
                x4 = shared[1]
                flags |= 64
        check(lib().ph_resnet_forward(plan.h, table, ptr(packed), ptr(x4 if x4 is not None else x), ptr(ws), ptr(f3), ptr(f4), flags,
                                      stream()), "ph_resnet_forward")
        ctx.x4 = x4
        ctx.net, ctx.plan, ctx.ws, ctx.packed, ctx.table = net, plan, ws, packed, table
        ctx.input_only = not net.training      # eval mode: the backward produces the image gradient only
        ctx.set_materialize_grads(False)
        ctx.nparams = len(params)
        return f3, f4

    @staticmethod
    def backward(ctx, g3, g4):
        net, plan = ctx.net, ctx.plan
        dev = ctx.ws.device
        if g4 is None:
            g4 = torch.zeros(plan.key[0], 512, device=dev, dtype=torch.float32)
        g4 = g4.contiguous().float()
        g3 = g3.contiguous().float() if g3 is not None else None
        if ctx.input_only:
            if not ctx.needs_input_grad[0]:
                ctx.ws = None
                return (None, None) + (None,) * ctx.nparams
            B, H, W = plan.key[0], plan.key[1], plan.key[2]
            dx = torch.empty(B, 3, H, W, device=dev, dtype=torch.float32)
            check(lib().ph_resnet_backward_input(plan.h, ctx.table, ptr(ctx.packed), ptr(ctx.ws), ptr(g3), ptr(g4), ptr(dx),
                                                 stream()), "ph_resnet_backward_input")
            ctx.ws = None
            return (dx, None) + (None,) * ctx.nparams
        grads, gptrs = net._alloc_trunk_grads()
        bp = ops.get_backward_precision()
        if plan.key[3] != ops.PREC_BF16:
            check(lib().ph_resnet_plan_set_backward_prec(plan.h, -1 if bp is None else bp), "ph_resnet_plan_set_backward_prec")
        # `net._no_bwd_overlap` (A/B and test switch): the whole backward on one stream instead of weight gradients on the
        # library's side stream (csrc/resnet_plan.hip: backward_impl) - same kernels, same BatchNorm gradients bitwise
        check(lib().ph_resnet_plan_set_backward_overlap(plan.h, 0 if getattr(net, "_no_bwd_overlap", False) else 1),
              "ph_resnet_plan_set_backward_overlap")
        hook = getattr(net, "_grad_ready_hook", None)
        if hook is None:
            check(lib().ph_resnet_backward(plan.h, ctx.table, ptr(ctx.packed), ptr(ctx.ws), ptr(g3), ptr(g4),
                                           ops.void_array(gptrs), stream()), "ph_resnet_backward")
        else:
            # data parallelism: after layers 4 and 3 the bulk of the gradient bytes is final; the hook starts their
            # all-reduce, which then overlaps the backward of layers 2, 1 and the stem
            ga = ops.void_array(gptrs)
            for part in (0, 1):
                check(lib().ph_resnet_backward_part(plan.h, ctx.table, ptr(ctx.packed), ptr(ctx.ws), ptr(g3), ptr(g4),
                                                    ga, part, stream()), "ph_resnet_backward_part")
                if part == 0:
                    hook()
        ctx.ws = None
        return (None, None) + tuple(grads)


def _x4_format(prec):
    """Layout of the packed NHWC4 input per arithmetic: 0 = bf16, 1 = fp32 (every split-plane mode), 2 = the two fp16
    planes of the half-pair mode (16 bytes per pixel, like fp32)."""
    return {ops.PREC_BF16: 0, ops.PREC_FP16X3: 2}.get(prec, 1)


def pack_shared_input(x, nets):
    """Pack the image batch `x` [B,3,H,W] f32 ONCE into the trunk's NHWC4 input layout for several networks that read it
    (the reference feeds the same x_path to the student and to the teacher, train_test_path_multi_distill.py:249,256, and
    each of its ResNets starts from the NCHW tensor).  The next forward of each net on this very tensor skips its own
    packing pass.  Must run on a stream the consumers are ordered after."""
    x = require_cuda(x, "x_path")
    if x.dtype != torch.float32 or not x.is_contiguous():
        return None
    B, Cc, H, W = x.shape
    prec = ops.get_precision()      # (every split-plane mode reads fp32 activations: one packed tensor serves bf16x6 and bf16x3)
    x4 = torch.empty(B, H, W, 4, device=x.device, dtype=torch.bfloat16 if prec == ops.PREC_BF16 else torch.float32)
    check(lib().ph_pack_input(ptr(x), ptr(x4), B, H, W, prec, stream()), "ph_pack_input")
    for net in nets:
        net._x4_shared = ((x.data_ptr(), tuple(x.shape), _x4_format(prec)), x4)
    return x4


class ResNet(nn.Module):
    """Same constructor / attributes / state_dict as the reference ResNet (resnets.py:126-191)."""

    def __init__(self, block, layers, path_dim=32, act=None, num_classes=7, return_grad="False",
                 zero_init_residual=False):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc_new1 = nn.Sequential(nn.Linear(512 * block.expansion, path_dim), nn.BatchNorm1d(path_dim),
                                     nn.ReLU(inplace=True))
        self.fc_new2 = nn.Linear(path_dim, num_classes)
        self.act = act
        self.return_grad = return_grad
        self.output_range = Parameter(torch.FloatTensor([6]), requires_grad=False)
        self.output_shift = Parameter(torch.FloatTensor([-3]), requires_grad=False)
        for m in self.modules():   # resnets.py:176-181
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)
        if list(layers) != [2, 2, 2, 2] or block is not BasicBlock:
            raise NotImplementedError("only ResNet-18 (BasicBlock, [2,2,2,2]) is on the hot path; the reference's "
                                      "ResNet34/50 factories are dead code (SURVEY.md section 2.1 row 3)")
        self._plans = {}
        self._packed = None
        self._packed_versions = None
        self._packed_key = None
        self._ws_cache = {}
        self._table = None
        self._table_key = None

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------ runtime plumbing
    def _units(self):
        """(conv, bn) pairs in the C-ABI unit order (include/pathomic_hip.h)."""
        u = [(self.conv1, self.bn1)]
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                u.append((blk.conv1, blk.bn1))
                u.append((blk.conv2, blk.bn2))
                if blk.downsample is not None:
                    u.append((blk.downsample[0], blk.downsample[1]))
        return u

    def _trunk_params(self):
        out = []
        for conv, bn in self._units():
            out += [conv.weight, bn.weight, bn.bias]
        return out

    def _param_table(self):
        ptrs = []
        for conv, bn in self._units():
            ptrs += [conv.weight.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                     bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()]
        key = tuple(ptrs)
        if key != self._table_key:
            self._table = ops.void_array(ptrs)
            self._table_key = key
        return self._table

    def _get_plan(self, B, H, W, prec=None):
        key = (B, H, W, ops.get_precision() if prec is None else prec)
        if key not in self._plans:
            self._plans[key] = _Plan(*key)
        return self._plans[key]

    def _get_packed(self, plan):
        # raw-pointer updates (fused Adam / EMA kernels) are invisible to torch's version counters: networks
        # that receive them follow the global weight epoch; frozen networks (the teacher) are packed once
        follow = getattr(self, "_follow_epoch", None)
        if follow is None:
            follow = any(c.weight.requires_grad for c, _ in self._units())
        vers = (ops.weight_epoch() if follow else -1, plan.key[3]) + tuple((c.weight._version, c.weight.data_ptr())
                                                                for c, _ in self._units())
        if self._packed is None or self._packed_versions != vers or self._packed.numel() != plan.packed_bytes:
            dev = self.conv1.weight.device
            if self._packed is None or self._packed.numel() != plan.packed_bytes:
                self._packed = torch.empty(plan.packed_bytes, device=dev, dtype=torch.uint8)
            check(lib().ph_resnet_pack_weights(plan.h, self._param_table(), ptr(self._packed), stream()),
                  "ph_resnet_pack_weights")
            self._packed_versions = vers
        return self._packed

    def _get_workspace(self, plan, persistent):
        dev = self.conv1.weight.device
        if persistent:
            return torch.empty(plan.ws_bytes, device=dev, dtype=torch.uint8)
        # A captured step graph holds the RAW pointer of the workspace it was captured with, so a workspace that is
        # requested WHILE a stream capture is running is pinned for good (releasing it when a forward of another shape
        # arrives - an evaluation batch between training epochs - would let later replays write into memory the
        # allocator has handed to someone else).  Everything else (ragged last batches, evaluation batches, the bench's
        # extra runs) lives in a small LRU: at B = 64 / 512 x 512 one workspace is 2.6 GB, and a long run with varying
        # batch sizes must not keep one per shape.  release_workspaces() frees explicitly.
        key = plan.key
        pinned = self.__dict__.setdefault("_ws_pinned", set())
        if torch.cuda.is_current_stream_capturing():
            pinned.add(key)
        ws = self._ws_cache.pop(key, None)
        if ws is None:
            loose = [k for k in self._ws_cache if k not in pinned]
            for k in loose[:max(0, len(loose) - (self._WS_LRU - 1))]:      # dict order = least recently used first
                del self._ws_cache[k]
            ws = torch.empty(plan.ws_bytes, device=dev, dtype=torch.uint8)
        elif not torch.cuda.is_current_stream_capturing():
            # the cached block goes back to the pool of the stream that allocated it when it is evicted: tell the allocator
            # about every other stream whose kernels read it (the forward-only networks run on side streams), so that an
            # eviction cannot recycle memory a side stream's kernels still use
            ws.record_stream(torch.cuda.current_stream())
        self._ws_cache[key] = ws                                          # (re-)inserted last = most recently used
        return ws

    _WS_LRU = 2      # un-pinned workspaces kept per network

    def pinned_workspaces(self):
        """The workspace tensors a stream capture has pinned (a captured graph holds their raw pointers): DistillStep keeps
        references to them next to the graph, so that release_workspaces() cannot return their memory to the allocator while
        the graph can still be replayed."""
        pinned = self.__dict__.get("_ws_pinned", set())
        return [ws for k, ws in self._ws_cache.items() if k in pinned]

    def release_workspaces(self, keep=None):
        """Free the cached trunk workspaces (all, or all but the plan key `keep` = (B, H, W, precision)), pinned ones
        included.  Only legal while no captured HIP graph replays this network: DistillStep.load_state_dict / a fresh
        enable_graph() rebuild theirs."""
        for k in [k for k in self._ws_cache if k != keep]:
            del self._ws_cache[k]
        self.__dict__.setdefault("_ws_pinned", set()).intersection_update({keep})

    def _alloc_trunk_grads(self):
        """Gradient destinations of ph_resnet_backward.  With `_direct_grad` (set by DistillStep, whose optimiser
        keeps zeroed flat `.grad` views) the kernels write straight into `p.grad` - every trunk parameter receives
        exactly one contribution per backward, so overwrite == accumulate-into-zero - and autograd gets None for
        it (saves ~60 tiny AccumulateGrad launches per step).  Otherwise fresh tensors are returned to autograd."""
        grads, gptrs = [], []
        direct = getattr(self, "_direct_grad", False)
        for p in self._trunk_params():
            if direct and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32:
                grads.append(None)
                gptrs.append(p.grad.data_ptr())
            else:
                g = torch.empty_like(p)
                grads.append(g)
                gptrs.append(g.data_ptr())
        return grads, gptrs

    # ------------------------------------------------------------------ reference API
    def _forward_impl(self, x):
        lin, bn = self.fc_new1[0], self.fc_new1[1]
        self._caller_grad_mode = torch.is_grad_enabled()
        if not self.training and torch.is_grad_enabled() and x.requires_grad:
            # eval-mode net with a gradient to the IMAGE (MIA-2023 superpixel attention, train_test_MT_SP_Masking.py:62-75)
            f3, f4 = _TrunkFn.apply(x, self, *self._trunk_params())
            h = ops.LinearFn.apply(f4, lin.weight, lin.bias)
            features = ops.BN1dEvalFn.apply(h, bn, True)
            hazard = ops.LinearFn.apply(features, self.fc_new2.weight, self.fc_new2.bias)
            pred = None
            if self.act is not None:
                if not isinstance(self.act, nn.LogSoftmax):
                    raise NotImplementedError("only act_type 'LSM' (grading task) is on the hot path")
                pred = ops.LogSoftmaxFn.apply(hazard)
            return f3, features, hazard, pred, None
        if not self.training:
            # eval mode (the reference's test(), train_test_path_multi_distill.py:409-431): BatchNorm uses the running
            # statistics.  Forward only.
            with torch.no_grad():
                f3, f4 = _TrunkFn.apply(x, self, *self._trunk_params())
                h = ops.linear_fwd(f4, lin.weight, lin.bias)
                features = ops.bn1d_eval(h, bn, relu=True)
                hazard = ops.linear_fwd(features, self.fc_new2.weight, self.fc_new2.bias)
                pred = None
                if self.act is not None:
                    if not isinstance(self.act, nn.LogSoftmax):
                        raise NotImplementedError("only act_type 'LSM' (grading task) is on the hot path")
                    pred = ops.LogSoftmaxFn.apply(hazard)
            return f3, features, hazard, pred, None
        f3, f4 = _TrunkFn.apply(x, self, *self._trunk_params())
        h = ops.LinearFn.apply(f4, lin.weight, lin.bias)
        features = ops.BN1dFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                    True, True)
        hazard = ops.LinearFn.apply(features, self.fc_new2.weight, self.fc_new2.bias)
        pred = None
        if self.act is not None:
            if isinstance(self.act, nn.LogSoftmax):
                pred = ops.LogSoftmaxFn.apply(hazard)
            else:
                raise NotImplementedError("only act_type 'LSM' (grading task) is on the hot path")
        return f3, features, hazard, pred, None

    def forward(self, **kwargs):
        return self._forward_impl(kwargs["x_path"])


def ResNet18(pretrained=True, progress=True, path_dim=32, act=None, num_classes=1, **kwargs):
    """resnets.py:287-295.  The reference loads ImageNet weights from a hard-coded relative path
    (resnets.py:278-282, strict=False); here `pretrained` loads the same file only if it exists."""
    model = ResNet(BasicBlock, [2, 2, 2, 2], path_dim, act, num_classes, **kwargs)
    if pretrained:
        import os
        path = "../pathomic_fusion_20211126/pretrained_resnet/resnet18-5c106cde.pth"
        if os.path.exists(path):
            model.load_state_dict(torch.load(path), strict=False)
    return model
