"""Drop-in for the reference's MICCAI-2022/networks_new.py: define_net / define_optimizer /
define_scheduler / define_reg / define_act_layer / define_bifusion, MaxNet, PathomicNet, get_resnet.
Same factories, constructor arguments, tuple returns and state_dict keys; compute through the C-ABI."""
import torch
import torch.nn as nn
import torch.optim.lr_scheduler as lr_scheduler
from torch.nn import Parameter

from . import ops
from .fusion import BilinearFusion
from .resnets import ResNet18
from .utils import init_net, init_max_weights


def define_net(opt, k, path_only=False, omic_only=False):
    """networks_new.py:53-77"""
    net = None
    act = define_act_layer(act_type=opt.act_type)
    init_max = True if opt.init_type == "max" else False
    if opt.mode == "path":
        net = get_resnet(path_dim=opt.path_dim, act=act, label_dim=opt.label_dim)
    elif opt.mode == "omic":
        net = MaxNet(input_dim=opt.input_size_omic, omic_dim=opt.omic_dim, dropout_rate=opt.dropout_rate, act=act,
                     label_dim=opt.label_dim, init_max=init_max)
    elif opt.mode == "pathomic":
        if not path_only and not omic_only:
            if opt.fusion_type == "mmdynamics":
                raise NotImplementedError("PathomicNet_dynamics needs the reference's absent MMDynamic module "
                                          "(networks_new.py:372-402): out of scope")
            net = PathomicNet(opt=opt, act=act, k=k)
        if path_only:
            net = get_resnet(path_dim=opt.path_dim, act=act, label_dim=opt.label_dim)
        if omic_only:
            net = MaxNet(input_dim=opt.input_size_omic, omic_dim=opt.omic_dim, dropout_rate=opt.dropout_rate,
                         act=act, label_dim=opt.label_dim, init_max=init_max)
    else:
        raise NotImplementedError("model [%s] is not implemented" % opt.mode)
    return init_net(net, opt.init_type, opt.init_gain, opt.gpu_ids)


def define_optimizer(opt, model):
    """networks_new.py:80-90.  'adam' returns the fused multi-tensor HIP Adam (same update rule and
    state_dict surface as torch.optim.Adam with L2-in-grad weight decay)."""
    from .train_step import FusedAdam, FusedAdagrad
    if opt.optimizer_type == "adam":
        return FusedAdam(model.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2), weight_decay=opt.weight_decay)
    if opt.optimizer_type == "adagrad":                                                    # :86-87
        return FusedAdagrad(model.parameters(), lr=opt.lr, weight_decay=opt.weight_decay, initial_accumulator_value=0.1)
    # ('adabound', :82-83, needs the third-party `adabound` package the reference imports; absent here and out of scope)
    raise NotImplementedError("initialization method [%s] is not implemented" % opt.optimizer_type)


def define_reg(opt, model):
    """networks_new.py:93-108: the L1 term the trainers scale with opt.lambda_reg (train_test_MT.py:209-217,
    train_test_path_multi_distill.py:312-313).  A device scalar with a gradient (0 for 'none')."""
    from . import utils as U
    if opt.reg_type == "none":
        return 0
    elif opt.reg_type == "path":
        return U.regularize_path_weights(model=model)
    elif opt.reg_type == "mm":
        return U.regularize_MM_weights(model=model)
    elif opt.reg_type == "all":
        return U.regularize_weights(model=model)
    elif opt.reg_type == "omic":
        return U.regularize_MM_omic(model=model)
    raise NotImplementedError("reg method [%s] is not implemented" % opt.reg_type)


def define_scheduler(opt, optimizer):
    """networks_new.py:111-129"""
    if opt.lr_policy == "linear":
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.niter) / float(opt.niter_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    elif opt.lr_policy == "exp":
        return lr_scheduler.ExponentialLR(optimizer, 0.1, last_epoch=-1)
    elif opt.lr_policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    elif opt.lr_policy == "plateau":                                                       # :120-121
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.2, threshold=0.01, patience=5)
    elif opt.lr_policy == "cosine":
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.niter, eta_min=0)
    elif opt.lr_policy == "onecycle":                                                      # :124-125
        # (cycles beta1 with the learning rate: the fused optimisers read lr AND betas from their parameter group at every
        # step and hand them to the kernel through device memory, so a captured step follows the schedule)
        return lr_scheduler.OneCycleLR(optimizer, max_lr=1e-3, epochs=opt.niter + opt.niter_decay, steps_per_epoch=200)
    # (the reference RETURNS the exception object here instead of raising it, :126-127: its trainer then fails at the first
    # scheduler.get_lr(); raising at construction is the loud form of the same failure)
    raise NotImplementedError("learning rate policy [%s] is not implemented" % opt.lr_policy)


def define_act_layer(act_type="Tanh"):
    """networks_new.py:132-145"""
    if act_type == "Tanh":
        return nn.Tanh()
    if act_type == "ReLU":
        return nn.ReLU()
    if act_type == "Sigmoid":
        return nn.Sigmoid()
    if act_type == "LSM":
        return nn.LogSoftmax(dim=1)
    if act_type == "none":
        return None
    raise NotImplementedError("activation layer [%s] is not found" % act_type)


def define_bifusion(fusion_type, skip=1, use_bilinear=1, gate1=1, gate2=1, dim1=32, dim2=32, scale_dim1=1,
                    scale_dim2=1, mmhid=32, dropout_rate=0.25):
    """networks_new.py:148-154"""
    if fusion_type == "pofusion":
        return BilinearFusion(skip=skip, use_bilinear=use_bilinear, gate1=gate1, gate2=gate2, dim1=dim1, dim2=dim2,
                              scale_dim1=scale_dim1, scale_dim2=scale_dim2, mmhid=mmhid, dropout_rate=dropout_rate)
    raise NotImplementedError("fusion type [%s] is not found" % fusion_type)


class MaxNet(nn.Module):
    """Genomic SNN (networks_new.py:182-251): 4 x (Linear-ELU-AlphaDropout), ReLU, Linear, act.
    forward(**kwargs) -> (features, out, pred, None).  Fused forward-only calls for the frozen stage-2 teacher; a taped
    path (ops.LinearActFn / DropoutFn / ReluFn) when a gradient is required (stage-1 training, row f-1)."""

    def __init__(self, input_dim=80, omic_dim=32, return_grad="False", dropout_rate=0.25, act=None, label_dim=1,
                 init_max=True):
        super().__init__()
        hidden = [64, 48, 32, 32]
        self.act = act
        self.return_grad = return_grad
        dims = [input_dim, hidden[0], hidden[1], hidden[2], omic_dim]
        self.encoder = nn.Sequential(*[
            nn.Sequential(nn.Linear(dims[i], dims[i + 1]), nn.ELU(), nn.AlphaDropout(p=dropout_rate, inplace=False))
            for i in range(4)])
        self.relu = nn.ReLU(inplace=False)
        self.classifier = nn.Sequential(nn.Linear(omic_dim, label_dim))
        if init_max:
            init_max_weights(self)
        self.output_range = Parameter(torch.FloatTensor([6]), requires_grad=False)
        self.output_shift = Parameter(torch.FloatTensor([-3]), requires_grad=False)
        self.dropout_rate = dropout_rate
        self.rng_seed = 0xA11CE
        self._rng_offset = 0
        self.register_buffer("rng_step", torch.zeros(1, dtype=torch.int64), persistent=False)

    def forward(self, **kwargs):
        x = kwargs["x_omic"]
        if self.return_grad == "True":
            raise NotImplementedError("return_grad needs the reference's absent my_utils.compute_gradients")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(x)     # stage-1 teacher training (row f-1): same arithmetic, taped
        h = x
        self._rng_offset = 0
        for i in range(4):
            lin = self.encoder[i][0]
            h = ops.linear_fwd(h, lin.weight, lin.bias, ops.ACT_ELU)
            if self.training and self.dropout_rate > 0:
                ops.dropout_(h, self.dropout_rate, self.rng_seed, self._rng_offset, self.rng_step, alpha=True)
                self._rng_offset += h.numel()
        if self.training and self.dropout_rate > 0:
            ops.counter_inc(self.rng_step)
        features = ops.eltwise(h, None, ops.EW_RELU)
        out = ops.linear_fwd(features, self.classifier[0].weight, self.classifier[0].bias)
        pred = None
        if self.act is not None:
            if not isinstance(self.act, nn.LogSoftmax):
                raise NotImplementedError("only act_type 'LSM' is on the hot path")
            pred = ops.LogSoftmaxFn.apply(out)
        return features, out, pred, None


def _maxnet_forward_autograd(self, x):
    h = x
    self._rng_offset = 0
    for i in range(4):
        lin = self.encoder[i][0]
        h = ops.LinearActFn.apply(h, lin.weight, lin.bias, ops.ACT_ELU)
        if self.training and self.dropout_rate > 0:
            h = ops.DropoutFn.apply(h, self.dropout_rate, self.rng_seed, self._rng_offset, self.rng_step, True)
            self._rng_offset += h.numel()
    if self.training and self.dropout_rate > 0:
        ops.counter_inc(self.rng_step)
    features = ops.ReluFn.apply(h)
    out = ops.LinearFn.apply(features, self.classifier[0].weight, self.classifier[0].bias)
    pred = None
    if self.act is not None:
        if not isinstance(self.act, nn.LogSoftmax):
            raise NotImplementedError("only act_type 'LSM' is on the hot path")
        pred = ops.LogSoftmaxFn.apply(out)
    return features, out, pred, None


MaxNet._forward_autograd = _maxnet_forward_autograd


def get_resnet(path_dim=32, act=None, label_dim=1, **kwargs):
    """networks_new.py:258-259"""
    return ResNet18(path_dim=path_dim, act=act, num_classes=label_dim, **kwargs)


class PathomicNet(nn.Module):
    """Multi-modal teacher (networks_new.py:267-353): path_net + omic_net + fusion + classifier; forward
    returns the reference's 11-tuple."""

    def __init__(self, opt, act, k):
        super().__init__()
        init_max = True if opt.init_type == "max" else False
        self.path_net = get_resnet(path_dim=opt.path_dim, act=act, label_dim=opt.label_dim, return_grad=opt.return_grad)
        self.omic_net = MaxNet(input_dim=opt.input_size_omic, omic_dim=opt.omic_dim, return_grad=opt.return_grad,
                               dropout_rate=opt.dropout_rate, act=act, label_dim=opt.label_dim, init_max=init_max)
        self.bilinear_dim = 20
        self.task = opt.task
        self.fusion = define_bifusion(fusion_type=opt.fusion_type, skip=opt.skip, use_bilinear=opt.use_bilinear,
                                      gate1=opt.path_gate, gate2=opt.omic_gate, dim1=opt.path_dim, dim2=opt.omic_dim,
                                      scale_dim1=opt.path_scale, scale_dim2=opt.omic_scale, mmhid=opt.mmhid,
                                      dropout_rate=opt.dropout_rate)
        self.classifier = nn.Sequential(nn.Linear(opt.mmhid, opt.label_dim))
        self.act = act
        self.return_grad = opt.return_grad
        self.cut_fuse_grad = opt.cut_fuse_grad
        self.fusion_type = opt.fusion_type
        self.output_range = Parameter(torch.FloatTensor([6]), requires_grad=False)
        self.output_shift = Parameter(torch.FloatTensor([-3]), requires_grad=False)

    def forward(self, **kwargs):
        # (the SNN first: its ~15 short launches are independent of the trunk; in front of it they are covered by the other
        # networks' kernels of the distillation step, behind it they would be a latency-bound tail before the fusion)
        omic_vec, hazard_omic, pred_omic, omic_grads = self.omic_net(x_omic=kwargs["x_omic"])
        path_vec_f3, path_vec, hazard_path, pred_path, path_grads = self.path_net(x_path=kwargs["x_path"])
        if self.fusion_type == "concat":
            raise NotImplementedError("fusion_type concat is not on the hot path (default pofusion)")
        if self.cut_fuse_grad:
            features = self.fusion(path_vec.detach(), omic_vec.detach())   # :302-306 (no copy: the fusion never writes its inputs)
        else:
            features = self.fusion(path_vec, omic_vec)
        hazard = ops.LinearFn.apply(features, self.classifier[0].weight, self.classifier[0].bias)
        pred = None
        if self.act is not None:
            if not isinstance(self.act, nn.LogSoftmax):
                raise NotImplementedError("only act_type 'LSM' is on the hot path")
            pred = ops.LogSoftmaxFn.apply(hazard)
        logits = [hazard_path, hazard_omic, hazard]
        return features, path_vec, omic_vec, path_vec_f3, logits, pred, pred_path, pred_omic, None, path_grads, omic_grads

    def __hasattr__(self, name):   # networks_new.py:356-369
        for d in ("_parameters", "_buffers", "_modules"):
            if d in self.__dict__ and name in self.__dict__[d]:
                return True
        return False
