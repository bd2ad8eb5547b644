"""Training-side operators on the hand-written kernels (`AlphaNNet.train`, alpha_nnet.py:58-59).

`SplitConv3x3`: the tower's 128 -> 128 3x3 convolution as a `torch.autograd.Function` whose forward pass and input
gradient run on `k_conv3x3_f16s` (csrc/conv_split.hip: float32 accuracy on the f16 matrix pipe, each operand split into
f16 hi + lo) -- the input gradient of a stride-1 'same' convolution is the same convolution with the taps flipped and the
channel axes swapped -- and whose weight gradient runs on `k_wgrad_f16s` (csrc/train_wgrad.hip; shapes whose LDS planes
do not fit fall back to the library's kernel).  Activations are channels-last in memory, which is the layout the kernels
read, so no tensor is permuted or copied.

The kernel scales its input by a power of two before splitting it (the f16 range is +-65504 and values far below the
largest one lose their low bits): the scale is taken from the tensor's largest magnitude on the device (2^11 <= max * scale
< 2^12), without a host round trip -- gradients of 1e-6 and activations of 10 both keep ~22 significant bits.
"""
import os

import torch

from ._lib import check, lib
from .net import F16S_TAIL_OFFSET, F16S_WEIGHT_BYTES


_NATIVE_WGRAD = os.environ.get("SNK_TRAIN_WGRAD", "native") != "torch"    # `torch`: the library's weight gradient (A/B runs)
_scratch = {}
# (data_ptr, numel, {., ., scale, 1 / scale}) of the tensor the LAST batch-norm kernel wrote: the convolution that reads it next
# (forward: the following layer; backward: this layer's own convolution) takes its input scale from there.  A live tensor's
# address cannot belong to another tensor, so an address + size match with the most recent producer identifies it.
_last_scale = None


def _partials(device):
    """the block-partials scratch of the two-stage reductions (one per device; stream-ordered reuse)"""
    key = (device.type, device.index)
    if key not in _scratch:
        _scratch[key] = torch.empty(lib().snk_bn_train_partials(), dtype=torch.float32, device=device)
    return _scratch[key]


def _unit(device):
    """identity batch-norm scale / shift for the convolution kernel's fused epilogue"""
    key = ("unit", device.type, device.index)
    if key not in _scratch:
        _scratch[key] = (torch.ones(128, device=device), torch.zeros(128, device=device))
    return _scratch[key]


def _conv_same(x_nhwc, k_hwio):
    """x: contiguous float32 cuda [n, h, w, 128]; k: float32 [3, 3, 128, 128] (kh, kw, cin, cout) -> [n, h, w, 128]"""
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    n, h, w, c = x_nhwc.shape
    image = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device=x_nhwc.device)
    check(L.snk_conv3x3_prepare_weights_f16s(k_hwio.contiguous().data_ptr(), image.data_ptr(), 1.0, st))
    global _last_scale
    tail = image[F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 16].view(torch.float32)
    if _last_scale is not None and _last_scale[0] == x_nhwc.data_ptr() and _last_scale[1] == x_nhwc.numel():
        tail[2:4] = _last_scale[2][2:4]        # the kernel that wrote this tensor a moment ago also took its largest magnitude
    else:
        check(L.snk_conv3x3_f16s_input_scale(x_nhwc.data_ptr(), x_nhwc.numel(), image.data_ptr(), _partials(x_nhwc.device).data_ptr(), st))
    _last_scale = None
    ones, zeros = _unit(x_nhwc.device)
    out = torch.empty_like(x_nhwc)
    check(L.snk_conv3x3_bn_f16s(x_nhwc.data_ptr(), image.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None, out.data_ptr(),
                                n, h, w, 0, st))
    return out, tail                                       # the tail carries the input's scale


def _input_scale(x_nhwc):
    """{., ., scale, 1 / scale} of a tensor that no convolution call has scaled yet"""
    image = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device=x_nhwc.device)
    check(lib().snk_conv3x3_f16s_input_scale(x_nhwc.data_ptr(), x_nhwc.numel(), image.data_ptr(), _partials(x_nhwc.device).data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
    return image[F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 16].view(torch.float32)


def _wgrad(x_nhwc, dy_nhwc, x_tail, dy_tail):
    """[3, 3, 128, 128] weight gradient on k_wgrad_f16s, or None when the shape does not fit its LDS planes"""
    L = lib()
    n, h, w, _ = x_nhwc.shape
    need = L.snk_conv3x3_wgrad_partials(h, w)
    if need <= 0:
        return None
    key = ("wgrad", x_nhwc.device.type, x_nhwc.device.index)
    if key not in _scratch or _scratch[key].numel() < need:
        _scratch[key] = torch.empty(need, dtype=torch.float32, device=x_nhwc.device)
    dk = torch.empty(3, 3, 128, 128, dtype=torch.float32, device=x_nhwc.device)
    check(L.snk_conv3x3_wgrad_f16s(x_nhwc.data_ptr(), dy_nhwc.data_ptr(), x_tail.data_ptr(), dy_tail.data_ptr(),
                                   _scratch[key].data_ptr(), dk.data_ptr(), n, h, w, torch.cuda.current_stream().cuda_stream))
    return dk


def usable(x, k):
    """the operator covers what the tower needs: float32 on the GPU, 3x3, 128 -> 128 channels"""
    return x.is_cuda and x.dtype == torch.float32 and tuple(k.shape) == (3, 3, 128, 128) and k.dtype == torch.float32


class SplitConv3x3(torch.autograd.Function):
    """y = conv2d(x, k), stride 1, 'same' zero padding.  x: [n, 128, h, w] (any strides; channels-last costs nothing),
    k: Keras layout [3, 3, cin, cout].  Returns a channels-last [n, 128, h, w] tensor."""

    @staticmethod
    def forward(ctx, x, k):
        x_nhwc = x.permute(0, 2, 3, 1).contiguous()            # a view when x is channels-last
        y, x_tail = _conv_same(x_nhwc, k)
        ctx.save_for_backward(x_nhwc, k, x_tail)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        x_nhwc, k, x_tail = ctx.saved_tensors
        dy_nhwc = dy.permute(0, 2, 3, 1).contiguous()
        dx = dk = dy_tail = None
        if ctx.needs_input_grad[0]:                            # the same convolution, taps flipped, channel axes swapped
            dx, dy_tail = _conv_same(dy_nhwc, k.flip(0, 1).permute(0, 1, 3, 2))
            dx = dx.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1] and _NATIVE_WGRAD:          # the reduction over pixels on k_wgrad_f16s
            dk = _wgrad(x_nhwc, dy_nhwc, x_tail, dy_tail if dy_tail is not None else _input_scale(dy_nhwc))
        if ctx.needs_input_grad[1] and dk is None:             # shapes whose planes do not fit the LDS: the library's kernel
            g = torch.ops.aten.convolution_backward(dy_nhwc.permute(0, 3, 1, 2), x_nhwc.permute(0, 3, 1, 2),
                                                    k.permute(3, 2, 0, 1), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                    [False, True, False])[1]
            dk = g.permute(2, 3, 1, 0)                         # (cout, cin, kh, kw) -> (kh, kw, cin, cout)
        return dx, dk


# ---------------------------------------------------------------------------------------------------------------------
BN_EPS = 1e-3           # Keras BatchNormalization default epsilon (alpha_nnet.py:23-46)


def bn_usable(y):
    return y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and y.shape[1] == 128


class FusedBatchNormAct(torch.autograd.Function):
    """out = act(gamma * (y - mean) / sqrt(var + eps) + beta (+ residual)), mean / biased variance over (N, H, W) of ALL
    ranks, on the kernels of csrc/train.hip: two passes over the activation forward, two backward.
    y, residual: [n, 128, h, w]; returns (out channels-last, mean, var, count) -- the last three for the moving averages.
    `dist` is torch.distributed when the step runs data-parallel (the 256 sums are all-reduced), else None."""

    @staticmethod
    def forward(ctx, y, gamma, beta, residual, relu, dist):
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        y_nhwc = y.permute(0, 2, 3, 1).contiguous()
        rows = y_nhwc.shape[0] * y_nhwc.shape[1] * y_nhwc.shape[2]
        c = 128
        stat = torch.empty(2 * c + 1, dtype=torch.float32, device=y.device)
        check(L.snk_bn_train_sums(y_nhwc.data_ptr(), rows, _partials(y.device).data_ptr(), stat.data_ptr(), st))
        stat[2 * c] = float(rows)
        if dist is not None:
            dist.all_reduce(stat)
        n = stat[2 * c]
        mean = stat[:c] / n
        var = (stat[c:2 * c] / n - mean * mean).clamp_min(0.0)
        inv = torch.rsqrt(var + BN_EPS)
        scale = (gamma * inv).contiguous()
        shift = (beta - mean * scale).contiguous()
        res_nhwc = residual.permute(0, 2, 3, 1).contiguous() if residual is not None else None
        global _last_scale
        out = torch.empty_like(y_nhwc)
        tail = torch.empty(4, dtype=torch.float32, device=y.device)
        check(L.snk_bn_train_apply(y_nhwc.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                   res_nhwc.data_ptr() if res_nhwc is not None else None, out.data_ptr(), rows, int(relu),
                                   _partials(y.device).data_ptr(), tail.data_ptr(), st))
        _last_scale = (out.data_ptr(), out.numel(), tail)
        ctx.save_for_backward(y_nhwc, out, gamma, mean.contiguous(), inv.contiguous(), n)
        ctx.relu, ctx.has_res, ctx.dist = bool(relu), residual is not None, dist
        res = out.permute(0, 3, 1, 2)
        ctx.mark_non_differentiable(mean, var, n)
        return res, mean, var, n

    @staticmethod
    def backward(ctx, dout, _dm, _dv, _dn):
        y_nhwc, out, gamma, mean, inv, n = ctx.saved_tensors
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        c = 128
        rows = y_nhwc.shape[0] * y_nhwc.shape[1] * y_nhwc.shape[2]
        d_nhwc = dout.permute(0, 2, 3, 1).contiguous()
        red = torch.empty(2 * c, dtype=torch.float32, device=dout.device)
        check(L.snk_bn_train_grad_sums(d_nhwc.data_ptr(), out.data_ptr(), y_nhwc.data_ptr(), mean.data_ptr(), inv.data_ptr(), rows,
                                       int(ctx.relu), _partials(dout.device).data_ptr(), red.data_ptr(), st))
        dbeta, dgamma = red[:c].clone(), red[c:].clone()                  # this rank's share of the parameter gradients
        if ctx.dist is not None:
            ctx.dist.all_reduce(red)                                      # the input gradient needs the global reductions
        a = (gamma * inv).contiguous()
        b = (red[:c] / n).contiguous()
        cc = (red[c:] / n).contiguous()
        dx = torch.empty_like(y_nhwc)
        g = torch.empty_like(y_nhwc) if ctx.has_res else None
        global _last_scale
        tail = torch.empty(4, dtype=torch.float32, device=dout.device)
        check(L.snk_bn_train_grad_apply(d_nhwc.data_ptr(), out.data_ptr(), y_nhwc.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                        a.data_ptr(), b.data_ptr(), cc.data_ptr(), dx.data_ptr(),
                                        g.data_ptr() if g is not None else None, rows, int(ctx.relu),
                                        _partials(dout.device).data_ptr(), tail.data_ptr(), st))
        _last_scale = (dx.data_ptr(), dx.numel(), tail)
        return dx.permute(0, 3, 1, 2), dgamma, dbeta, (g.permute(0, 3, 1, 2) if g is not None else None), None, None
