"""torch.autograd glue over the C-ABI: every Function's forward/backward is a call into
libpathomic_hip.so on raw device pointers + the current HIP stream.  torch supplies memory and the
autograd tape only; there is no eager-PyTorch compute fallback."""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check, ptr, stream, require_cuda

ACT_NONE, ACT_RELU, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3
EW_RELU, EW_GATE, EW_RELU_BWD, EW_ADD, EW_ELU_BWD, EW_MUL = 0, 1, 2, 3, 4, 5
PREC_BF16, PREC_BF16X6, PREC_BF16X3, PREC_FP16X3, PREC_FP16X1 = 0, 1, 2, 3, 4

_precision = PREC_BF16
_fwd_only_precision = None      # optional other arithmetic for forward-only (no_grad) trunk forwards, see set_precision
_backward_precision = None      # optional other arithmetic for the backward's dgrad / wgrad kernels
_weight_epoch = 0


def bump_weight_epoch():
    """Call after parameters were modified through raw device pointers (fused Adam / EMA kernels), so that
    cached MFMA weight packings are refreshed."""
    global _weight_epoch
    _weight_epoch += 1


def weight_epoch():
    return _weight_epoch


def set_precision(mode):
    """Arithmetic of the ResNet trunk:
      'bf16'      perf mode: bf16 operands and activations, fp32 accumulation and statistics;
      'bf16x6'    parity mode: fp32 activations, operands split into three bf16 planes, six MFMA products (fp32-equivalent);
      'bf16x3'    fp32 activations, the three leading products only (16-bit operands, ~2^-16 per product): half the matrix work;
      'fp16x3'    half-pair mode: every tensor a convolution reads (activations, BatchNorm-backward dz under a per-tensor
                  power-of-two scale) is stored as two fp16 planes x = hi + lo * 2^-11 (22 significant bits, 4 B per element),
                  conv outputs and gradients stay fp32, three fp16 MFMA products per k-step (hi*hi, hi*lo, lo*hi: ~2^-22 per
                  product, fp32 accumulation) - the cheapest arithmetic that meets the 1e-3 parity tolerance everywhere;
      'fp16x3/x1' 'fp16x3' for every forward, ONE fp16 product (the hi planes) in the backward's dgrad / wgrad kernels;
      'bf16x6/x3' parity mode for every forward, three products in the backward's dgrad / wgrad kernels;
      'bf16x6+x3' parity mode for every forward that is followed by a backward (the student), 'bf16x3' for the forward-only
                  networks (the no_grad EMA / teacher forwards of train_test_path_multi_distill.py:253-256)."""
    global _precision, _fwd_only_precision, _backward_precision
    _backward_precision = None
    names = {"bf16": PREC_BF16, "bf16x6": PREC_BF16X6, "bf16x3": PREC_BF16X3, "fp16x3": PREC_FP16X3, PREC_BF16: PREC_BF16,
             PREC_BF16X6: PREC_BF16X6, PREC_BF16X3: PREC_BF16X3, PREC_FP16X3: PREC_FP16X3}
    if mode == "bf16x6+x3":
        _precision, _fwd_only_precision = PREC_BF16X6, PREC_BF16X3
    elif mode == "fp16x3/x1":
        # half-pair forward (logits, losses, GK-Refine weights as in 'fp16x3'); the backward's dgrad / wgrad multiply the hi
        # planes alone: one fp16 product on 11-bit operands with fp32 accumulation (gradients at ~1e-4 of their scale)
        _precision, _fwd_only_precision, _backward_precision = PREC_FP16X3, None, PREC_FP16X1
    elif mode == "bf16x6/x3":
        # every FORWARD in parity arithmetic (logits, losses, GK-Refine weights as in 'bf16x6'), the backward's dgrad /
        # wgrad kernels with three products
        _precision, _fwd_only_precision, _backward_precision = PREC_BF16X6, None, PREC_BF16X3
    else:
        _precision, _fwd_only_precision = names[mode], None


def get_backward_precision():
    return _backward_precision


def get_precision(fwd_only=False):
    """The trunk arithmetic in force (for a forward-only forward when `fwd_only`)."""
    return _fwd_only_precision if (fwd_only and _fwd_only_precision is not None) else _precision


def _f32(t):
    t = require_cuda(t)
    if t.dtype != torch.float32:
        raise RuntimeError("expected float32 tensor")
    return t.contiguous()


_ones_cache = {}


def _ones(n, device):
    key = (n, device)
    if key not in _ones_cache:
        _ones_cache[key] = torch.ones(n, device=device, dtype=torch.float32)
    return _ones_cache[key]


def sgemm(A, Bm, bias, out, M, N, K, sam, sak, sbk, sbn, act=ACT_NONE, accumulate=False):
    check(lib().ph_sgemm(ptr(A), ptr(Bm), ptr(bias), ptr(out), M, N, K, sam, sak, sbk, sbn, N, act,
                         int(accumulate), stream()), "ph_sgemm")
    return out


def linear_fwd(x, w, b, act=ACT_NONE):
    """y[B,N] = act(x[B,K] @ w[N,K]^T + b)   (nn.Linear)"""
    x, w = _f32(x), _f32(w)
    Bn, K = x.shape
    N = w.shape[0]
    y = torch.empty(Bn, N, device=x.device, dtype=torch.float32)
    if K >= 4096 and Bn * N <= 256 * 256:
        nsplit = 128 if K >= 8192 else 32     # 64-256 workgroups: a 16641-deep K in 5 LDS steps per workgroup instead of 17
        part = torch.empty(nsplit * Bn * N, device=x.device, dtype=torch.float32)
        check(lib().ph_sgemm_splitk(ptr(x), ptr(w), ptr(b), ptr(y), ptr(part), nsplit, Bn, N, K, K, 1, 1, K, N, act,
                                    stream()), "ph_sgemm_splitk")
        return y
    return sgemm(x, w, b, y, Bn, N, K, K, 1, 1, K, act)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return linear_fwd(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _f32(g)
        Bn, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            sgemm(g, w, None, dx, Bn, K, N, N, 1, K, 1)              # dX = dY @ W
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            sgemm(g, x, None, dw, N, K, Bn, 1, N, K, 1)              # dW = dY^T @ X
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(N, device=g.device, dtype=torch.float32)
            sgemm(_ones(Bn, g.device), g, None, db, 1, N, Bn, 0, 1, N, 1)   # db = 1^T dY
        return dx, dw, db


# ---- autograd forms of the SNN / fusion operators (stage-1 teacher training).  The frozen teacher of the
# stage-2 hot path keeps its fused forward-only calls; these are used when a gradient is required.
class LinearActFn(torch.autograd.Function):
    """y = act(x @ w^T + b) with act in {none, relu, elu} fused into the GEMM epilogue; backward from the saved y."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        y = linear_fwd(x, w, b, act)
        ctx.save_for_backward(x, w, y)
        ctx.act, ctx.has_bias = act, b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = _f32(g)
        if ctx.act == ACT_RELU:
            g = eltwise(g, y, EW_RELU_BWD)
        elif ctx.act == ACT_ELU:
            g = eltwise(g, y, EW_ELU_BWD)
        elif ctx.act != ACT_NONE:
            raise NotImplementedError("LinearActFn backward: activation %d" % ctx.act)
        Bn, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            sgemm(g, w, None, dx, Bn, K, N, N, 1, K, 1)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            sgemm(g, x, None, dw, N, K, Bn, 1, N, K, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(N, device=g.device, dtype=torch.float32)
            sgemm(_ones(Bn, g.device), g, None, db, 1, N, Bn, 0, 1, N, 1)
        return dx, dw, db, None


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = eltwise(x, None, EW_RELU)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return eltwise(_f32(g), y, EW_RELU_BWD)


class DropoutFn(torch.autograd.Function):
    """(Alpha-)dropout with the counter-based RNG of ph_dropout_dev; the backward re-creates the mask from the step
    counter value of the forward call (a saved 8-byte device copy), so no mask tensor is stored."""

    @staticmethod
    def forward(ctx, x, p, seed, site_offset, step_counter, alpha):
        x = _f32(x)
        y = torch.empty_like(x)
        check(lib().ph_dropout_dev_to(ptr(x), ptr(y), y.numel(), p, seed, site_offset, ptr(step_counter), int(alpha), stream()),
              "ph_dropout_dev_to")
        if ctx.needs_input_grad[0]:      # (the frozen teacher's dropouts never run a backward: no 8-byte copy kernel for them)
            ctx.save_for_backward(step_counter.clone())
        ctx.args = (p, seed, site_offset, int(alpha))
        return y

    @staticmethod
    def backward(ctx, g):
        (ctr,) = ctx.saved_tensors
        p, seed, site_offset, alpha = ctx.args
        g = _f32(g)
        d = torch.empty_like(g)
        check(lib().ph_dropout_bwd_dev_to(ptr(g), ptr(d), d.numel(), p, seed, site_offset, ptr(ctr), alpha, stream()),
              "ph_dropout_bwd_dev_to")
        return d, None, None, None, None, None


class GateFn(torch.autograd.Function):
    """y = sigmoid(z) * h  (fusion.py:44-45)"""

    @staticmethod
    def forward(ctx, z, h):
        z, h = _f32(z), _f32(h)
        ctx.save_for_backward(z, h)
        return eltwise(z, h, EW_GATE)

    @staticmethod
    def backward(ctx, g):
        z, h = ctx.saved_tensors
        g = _f32(g)
        dz, dh = torch.empty_like(z), torch.empty_like(h)
        check(lib().ph_gate_bwd(ptr(g), ptr(z), ptr(h), ptr(dz), ptr(dh), z.numel(), stream()), "ph_gate_bwd")
        return dz, dh


class OuterFn(torch.autograd.Function):
    """o12[b] = vec(o1e[b] (x) o2e[b]) with an optional appended 1 on both operands (fusion.py:43,56-58)"""

    @staticmethod
    def forward(ctx, o1, o2, append_one):
        o1, o2 = _f32(o1), _f32(o2)
        ctx.save_for_backward(o1, o2)
        ctx.append_one = int(append_one)
        return outer(o1, o2, ctx.append_one)

    @staticmethod
    def backward(ctx, g):
        o1, o2 = ctx.saved_tensors
        g = _f32(g)
        d1, d2 = torch.empty_like(o1), torch.empty_like(o2)
        check(lib().ph_outer_bwd(ptr(g), ptr(o1), ptr(o2), ptr(d1), ptr(d2), o1.shape[0], o1.shape[1], o2.shape[1],
                                 ctx.append_one, stream()), "ph_outer_bwd")
        return d1, d2, None


class BN1dFn(torch.autograd.Function):
    """nn.BatchNorm1d (training statistics) optionally fused with ReLU."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, relu, update_running):
        x = _f32(x)
        Bn, Cn = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(Cn, device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        check(lib().ph_bn1d_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(invstd),
                                ptr(rm) if update_running else None, ptr(rv), ptr(nbt), Bn, Cn, 1e-5, 0.1, int(relu),
                                stream()), "ph_bn1d_fwd")
        ctx.save_for_backward(x, y, mean, invstd, gamma)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, mean, invstd, gamma = ctx.saved_tensors
        g = _f32(g)
        Bn, Cn = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty_like(gamma)
        db = torch.empty_like(gamma)
        check(lib().ph_bn1d_bwd(ptr(g), ptr(y), ptr(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(dx), ptr(dg), ptr(db),
                                Bn, Cn, int(ctx.relu), stream()), "ph_bn1d_bwd")
        return dx, dg, db, None, None, None, None, None


def bn1d_eval(x, bn, relu):
    """nn.BatchNorm1d in eval mode (running statistics), optionally + ReLU; forward only."""
    x = _f32(x)
    y = torch.empty_like(x)
    check(lib().ph_bn1d_eval(ptr(x), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), ptr(y),
                             x.shape[0], x.shape[1], bn.eps, int(relu), stream()), "ph_bn1d_eval")
    return y


class BN1dEvalFn(torch.autograd.Function):
    """bn1d_eval with a backward to the input (eval-mode BatchNorm1d is a fixed per-channel scale): used when a
    gradient flows through an eval-mode network to its inputs (MIA-2023 superpixel attention masks)."""

    @staticmethod
    def forward(ctx, x, bn, relu):
        y = bn1d_eval(x, bn, relu)
        ctx.save_for_backward(y)
        ctx.bn, ctx.relu = bn, relu
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        bn = ctx.bn
        g = _f32(g)
        dx = torch.empty_like(y)
        check(lib().ph_bn1d_eval_bwd(ptr(g), ptr(y), ptr(bn.weight), ptr(bn.running_var), ptr(dx), y.shape[0], y.shape[1],
                                     bn.eps, int(ctx.relu), stream()), "ph_bn1d_eval_bwd")
        return dx, None, None


class LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty_like(x)
        check(lib().ph_log_softmax(ptr(x), ptr(y), x.shape[0], x.shape[1], stream()), "ph_log_softmax")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        g = _f32(g)
        dx = torch.empty_like(y)
        check(lib().ph_log_softmax_bwd(ptr(g), ptr(y), ptr(dx), y.shape[0], y.shape[1], stream()), "ph_log_softmax_bwd")
        return dx


class NLLFn(torch.autograd.Function):
    """F.nll_loss(pred, grade) with mean reduction over `bnorm` samples (the global batch under DDP)."""

    @staticmethod
    def forward(ctx, pred, grade, bnorm):
        pred = _f32(pred)
        grade = require_cuda(grade).contiguous()
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        check(lib().ph_nll_fwd(ptr(pred), ptr(grade), ptr(loss), pred.shape[0], pred.shape[1], 1.0 / bnorm, stream()),
              "ph_nll_fwd")
        ctx.save_for_backward(grade)
        ctx.shape = tuple(pred.shape)
        ctx.bnorm = bnorm
        return loss

    @staticmethod
    def backward(ctx, g):
        grade, = ctx.saved_tensors
        g = _f32(g)
        d = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        check(lib().ph_nll_bwd(ptr(g), ptr(grade), ptr(d), ctx.shape[0], ctx.shape[1], 1.0 / ctx.bnorm, stream()),
              "ph_nll_bwd")
        return d, None, None


class KLFn(torch.autograd.Function):
    """DistillKL (reference KD_loss.py:13-17); gradient w.r.t. y_s only (y_t is detached on the hot path)."""

    @staticmethod
    def forward(ctx, y_s, y_t, T, bnorm):
        y_s, y_t = _f32(y_s), _f32(y_t)
        loss = torch.empty((), device=y_s.device, dtype=torch.float32)
        check(lib().ph_kl_fwd(ptr(y_s), ptr(y_t), ptr(loss), y_s.shape[0], y_s.shape[1], T, 1.0 / bnorm, stream()),
              "ph_kl_fwd")
        ctx.save_for_backward(y_s, y_t)
        ctx.T, ctx.bnorm = T, bnorm
        return loss

    @staticmethod
    def backward(ctx, g):
        y_s, y_t = ctx.saved_tensors
        g = _f32(g)
        d = torch.empty_like(y_s)
        check(lib().ph_kl_bwd(ptr(g), ptr(y_s), ptr(y_t), ptr(d), y_s.shape[0], y_s.shape[1], ctx.T, 1.0 / ctx.bnorm,
                              stream()), "ph_kl_bwd")
        return d, None, None, None


class L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty_like(x)
        nrm = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        check(lib().ph_l2norm_fwd(ptr(x), ptr(y), ptr(nrm), x.shape[0], x.shape[1], stream()), "ph_l2norm_fwd")
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, g):
        y, nrm = ctx.saved_tensors
        g = _f32(g)
        dx = torch.empty_like(y)
        check(lib().ph_l2norm_bwd(ptr(g), ptr(y), ptr(nrm), ptr(dx), y.shape[0], y.shape[1], stream()), "ph_l2norm_bwd")
        return dx


def eltwise(a, b, op):
    a = _f32(a)
    out = torch.empty_like(a)
    check(lib().ph_eltwise(ptr(a), ptr(b), ptr(out), a.numel(), op, stream()), "ph_eltwise")
    return out


def dropout_(x, p, seed, site_offset, step_counter, alpha=False):
    """In-place (alpha-)dropout; `step_counter` is a device uint64 advanced once per forward so that the launch
    is replayable inside a captured HIP graph."""
    if p > 0:
        check(lib().ph_dropout_dev(ptr(x), x.numel(), p, seed, site_offset, ptr(step_counter), int(alpha), stream()),
              "ph_dropout_dev")
    return x


def counter_inc(counter):
    check(lib().ph_counter_inc(ptr(counter), stream()), "ph_counter_inc")


def outer(o1, o2, append_one):
    o1, o2 = _f32(o1), _f32(o2)
    Bn, D1 = o1.shape
    D2 = o2.shape[1]
    out = torch.empty(Bn, (D1 + append_one) * (D2 + append_one), device=o1.device, dtype=torch.float32)
    check(lib().ph_outer(ptr(o1), ptr(o2), ptr(out), Bn, D1, D2, int(append_one), stream()), "ph_outer")
    return out


def void_array(ptrs):
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


# ---------------------------------------------------------------- L1 weight regulariser (define_reg)
_flat_registry = {}     # untyped-storage pointer of a FlatParams buffer -> weakref to it (train_step.FlatParams registers)


def register_flat(fp):
    import weakref
    _flat_registry[fp.flat.untyped_storage().data_ptr()] = weakref.ref(fp)


def _l1_segments(tensors):
    """[(param pointer, element count, gradient pointer or None, tensors covered)]: runs of parameters that sit back
    to back in one FlatParams buffer (16-byte aligned, zero padding in between: |0| = 0, sgn(0) = 0) become ONE
    segment whose gradient goes straight into the flat gradient buffer; anything else is a segment of its own."""
    segs = []
    for t in tensors:
        t = require_cuda(t, "parameter")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("define_reg expects contiguous float32 parameters")
        ref = _flat_registry.get(t.untyped_storage().data_ptr())
        fp = ref() if ref is not None else None
        if fp is not None and fp.flat.untyped_storage().data_ptr() == t.untyped_storage().data_ptr():
            off = (t.data_ptr() - fp.flat.data_ptr()) // 4
            end = off + (t.numel() + 3) // 4 * 4
            direct = (fp.grad is not None and t.requires_grad and t.grad is not None
                      and t.grad.data_ptr() == fp.grad.data_ptr() + 4 * off)
            last = segs[-1] if segs else None
            if (last is not None and last["fp"] is fp and last["end"] == off and last["direct"] == direct):
                last["end"] = end
                last["tensors"].append(t)
                continue
            segs.append(dict(fp=fp, off=off, end=end, direct=direct, tensors=[t]))
        else:
            segs.append(dict(fp=None, off=0, end=t.numel(), direct=False, tensors=[t]))
    return segs


class L1RegFn(torch.autograd.Function):
    """sum_i |W_i|.sum() of the reference's regularize_* helpers (utils.py:60-198) as a device scalar.  Backward:
    sgn(W) * upstream, accumulated IN PLACE into the flat gradient buffer where the parameter's .grad is a view of it
    (the optimiser's zero_grad() runs before backward() in every step class), returned to autograd otherwise."""

    @staticmethod
    def forward(ctx, *tensors):
        segs = _l1_segments(tensors)
        dev = tensors[0].device
        out = torch.zeros(1, device=dev, dtype=torch.float32)
        scratch = torch.empty(1024, device=dev, dtype=torch.float32)
        for sg in segs:
            w = sg["fp"].flat[sg["off"]:sg["end"]] if sg["fp"] is not None else sg["tensors"][0].reshape(-1)
            check(lib().ph_l1_sum(ptr(w), w.numel(), ptr(scratch), ptr(out), 1, stream()), "ph_l1_sum")
        ctx.segs = segs
        ctx.tensors = tensors
        return out[0]

    @staticmethod
    def backward(ctx, g):
        g = g.reshape(1).float().contiguous()
        grads = {}
        for sg in ctx.segs:
            if sg["direct"]:
                fp = sg["fp"]
                check(lib().ph_l1_sign_axpy(ptr(fp.flat[sg["off"]:sg["end"]]), ptr(fp.grad[sg["off"]:sg["end"]]),
                                            sg["end"] - sg["off"], ptr(g), 1.0, stream()), "ph_l1_sign_axpy")
                continue
            for t in sg["tensors"]:
                if not t.requires_grad:
                    continue
                d = torch.zeros_like(t)
                check(lib().ph_l1_sign_axpy(ptr(t), ptr(d), t.numel(), ptr(g), 1.0, stream()), "ph_l1_sign_axpy")
                grads[id(t)] = d
        return tuple(grads.get(id(t)) for t in ctx.tensors)


def l1_norm_sum(tensors):
    tensors = list(tensors)
    if not tensors:
        return None        # the reference's helpers return None when nothing matched (the caller's `lambda_reg * None` raises)
    return L1RegFn.apply(*tensors)
