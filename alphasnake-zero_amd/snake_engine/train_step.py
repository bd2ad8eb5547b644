"""One optimizer step of ``AlphaNNet.train`` (alpha_nnet.py:58-59: Keras ``fit`` = forward, backward, Adam) sequenced by hand
on the kernels of this library -- no autograd graph, no library convolution or GEMM.

Data flow of a step on ``n`` rows (h x w x 3 observations, 3 targets); every activation is channels-last float32
``[n * h * w][128]`` and stays in HBM between kernels:

  forward   stem          snk_stem_conv_f32                     y0
            batch norm    snk_bn_train_sums_f64 -> (all-reduce) -> snk_bn_train_finalize -> snk_bn_train_apply    out0 (+ its
                          power-of-two range for the convolution that reads it)
            tower layer   snk_conv3x3_prepare_weights_f16s_train + snk_conv3x3_f16s_stats (bare convolution; its epilogue also takes
                          the batch-norm sums) + batch norm (+ shortcut) + ReLU      y_l, out_l
            head          snk_head_conv1x1_sums -> finalize (1 channel) -> snk_head_dense_train_fwd      z, h, d1, q, sum (q - t)^2
  backward  head          snk_head_dense_train_bwd -> (all-reduce) -> snk_bn_train_grad_finalize -> snk_head_conv1x1_bwd
            tower layer   snk_bn_train_grad_sums_f64 -> (all-reduce) -> snk_bn_train_grad_finalize -> snk_bn_train_grad_apply,
                          snk_conv3x3_wgrad_f16s (weight gradient, straight into the flat gradient buffer),
                          snk_conv3x3_bn_f16s with the mirrored kernel (input gradient; a block's first layer adds the
                          shortcut's gradient in the same kernel's epilogue)
            stem          batch-norm backward + snk_stem_wgrad_f32
  update    (all-reduce of the ONE flat gradient buffer) -> snk_l2_sum (the regularization loss) -> snk_adam_l2_step

Parameters, gradients and Adam moments are single flat buffers; every tensor of the Keras weight list is a view.  With
``torch.distributed`` initialised the batch-norm sums (float64) and the gradient buffer are all-reduced: the result equals
the one-rank full-batch step up to float32 rounding.  A step whose learning rate is 0 (alpha_nnet.py:79-84: every step
after the 100th) can only move the batch-norm moving averages: ``forward_only`` runs just the forward half.

The deferred batch norm (round 5; ``SNK_TRAIN_DEFER_BN=0`` turns it off): the activation between the two convolutions of a
residual block -- out_l of an odd layer l, relu(bn(y_l)), no shortcut -- is read by the block's second convolution, by that
layer's weight gradient and, as a sign, by the batch-norm backward of layer l.  All three take it from y_l instead, as
relu(y_l * scale_l + shift_l) evaluated the way snk_bn_train_apply evaluates it (bit for bit the same values), while they stage
or read y_l: out_l and its mask bytes are never written nor read -- one element-wise pass over two 462 MB tensors less per
block in EVERY step, the forward-only ones included.  The range the second convolution scales its input by, which
snk_bn_train_apply measures while it writes, comes from the per-channel maxima the first convolution's epilogue takes next
to its sums (snk_conv3x3_f16s_stats_deferred -> snk_bn_train_finalize_range).

The shortcut's gradient (round 5; ``SNK_TRAIN_RES_MASK=0`` turns it off): the gradient a residual block hands to its shortcut is
the gradient at the block's output where that output's ReLU let the value through.  The batch-norm backward of the block's second
layer no longer writes that masked copy (462 MB per block): the gradient it READ stays where it is (the block's two
input-gradient launches alternate between two buffers), and the input-gradient launch of the block's first layer, which adds
the shortcut's gradient in its epilogue, takes it from there through the output's mask bytes
(snk_conv3x3_f16s_igrad_stats_masked_res), in place.

Weight images (round 5; ``SNK_TRAIN_BATCH_PREP=0`` turns it off): all tower layers' forward and input-gradient images are made in
two launches per optimizer step (snk_conv3x3_prepare_weights_f16s_train_batch) and kept while the weights stay -- the
forward-only steps make none.  An image's input scale is written straight into its tail by the kernel that writes the
convolution's input (``tail_out[l]`` is a view of layer l + 1's forward image, ``tail_dy[l]`` of layer l's input-gradient image).
"""
import ctypes
import os

import numpy as np
import torch

from ._lib import check, lib
from .net import F16S_TAIL_OFFSET, F16S_WEIGHT_BYTES

_CONV_STATS = os.environ.get("SNK_TRAIN_CONV_STATS", "1") != "0"      # 0: batch-norm sums in a pass of their own (A/B runs)
_IGRAD_STATS = os.environ.get("SNK_TRAIN_IGRAD_STATS", "1") != "0"    # 0: the batch-norm BACKWARD sums in a pass of their own
_BATCH_PREP = os.environ.get("SNK_TRAIN_BATCH_PREP", "1") != "0"      # 0: every layer's weight images made one by one, each step (A/B runs)
_RES_MASK = os.environ.get("SNK_TRAIN_RES_MASK", "1") != "0"          # 0: the shortcut's gradient is written as a masked copy (A/B runs)
_HEAD_FUSED = os.environ.get("SNK_TRAIN_HEAD_FUSED", "1") != "0"      # 0: the head's 1x1 convolution in a pass of its own (A/B runs)
_DEFER_STEM = os.environ.get("SNK_TRAIN_DEFER_STEM", "1") != "0"      # 0: the stem's batch norm + ReLU output is written (A/B runs)
_DEFER_BN = os.environ.get("SNK_TRAIN_DEFER_BN", "1") != "0"          # 0: every layer's batch norm + ReLU output is written (A/B runs)
BN_EPS, BN_MOMENTUM, L2_C = 1e-3, 0.99, 1e-5
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7
C = 128


def _p(t):
    return None if t is None else t.data_ptr()


def supported(input_shape, blocks=1):
    """the shapes whose weight-gradient kernels exist (square observations that fit their LDS images).  A net WITHOUT residual
    blocks (stem -> head; not a reference shape, alpha_nnet.py:25 builds four) has no tower for the batched weight images, the
    fused head sums and the deferred stem to hang on: it is not supported and fit() says so."""
    h, w = int(input_shape[0]), int(input_shape[1])
    L = lib()
    return blocks >= 1 and h == w and L.snk_conv3x3_wgrad_partials(h, w) > 0 and L.snk_stem_wgrad_partials(1, h, w) > 0


class TrainStep:
    def __init__(self, weights, input_shape, max_rows, device, dist=None):
        self.L = lib()
        self.dev = torch.device(device)
        self.dist = dist
        self.h, self.w = int(input_shape[0]), int(input_shape[1])
        self.hw = self.h * self.w
        self.blocks = (len(weights) - 14) // 10
        self.n_layers = 1 + 2 * self.blocks                     # stem + tower layers, all 128 channels wide
        self.max_rows = int(max_rows)
        if not supported(input_shape, self.blocks):
            raise ValueError(f"TrainStep: no weight-gradient kernel for {self.h} x {self.w} observations and {self.blocks} residual blocks")
        ws = [np.asarray(w, np.float32) for w in weights]
        # ---- flat parameters: per conv layer kernel, gamma, beta; then Dense kernels and biases (Keras order without the moving stats)
        self.param_idx, self.kernel_idx = [], []
        i = 0
        for _ in range(self.n_layers + 1):                      # + the head's 1x1 convolution
            self.param_idx += [i, i + 1, i + 2]
            self.kernel_idx.append(i)
            i += 5
        self.param_idx += [i, i + 1, i + 2, i + 3]
        self.kernel_idx += [i, i + 2]
        self.fc_idx = i
        sizes = [ws[j].size for j in self.param_idx]
        self.n_params = int(sum(sizes))
        f = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.dev)
        self.W, self.G, self.M, self.V = f(self.n_params), f(self.n_params + 1), f(self.n_params), f(self.n_params)
        self.decay = f(self.n_params, torch.uint8)
        self.view, self.gview, self.shape = {}, {}, {}
        o = 0
        for j, sz in zip(self.param_idx, sizes):
            self.view[j], self.gview[j], self.shape[j] = self.W[o:o + sz], self.G[o:o + sz], ws[j].shape
            self.view[j].copy_(torch.as_tensor(ws[j].reshape(-1)))
            if j in self.kernel_idx:
                self.decay[o:o + sz] = 1
            o += sz
        # a block's first layers (odd l) whose batch norm + ReLU is applied by the kernels that read its output (see the module text)
        self.defer = bool(_DEFER_BN and _CONV_STATS and self.L.snk_train_deferred_bn_supported(self.h, self.w) == 1)
        # ... and the stem's (layer 0; readers: the first tower convolution, its weight gradient, the first block's shortcut, the stem's
        # own batch-norm backward) -- only together with the masked shortcut gradient, whose launch is the one that takes layer 0's sums
        self.defer_stem = bool(self.defer and _DEFER_STEM and _RES_MASK and _IGRAD_STATS)
        self.moving = {j: torch.as_tensor(ws[j].reshape(-1).copy(), device=self.dev) for j in range(len(ws)) if j not in self.view}
        self.adam_t = 0
        # ---- activations kept for the backward pass, gradients in flight
        act = self.max_rows * self.hw * C
        self.y = [f(act) for _ in range(self.n_layers)]
        self.out = [None if self._deferred(l) else f(act) for l in range(self.n_layers)]
        self.res_mask = bool(_RES_MASK and _IGRAD_STATS)
        self.dA, self.dY, self.gres = f(act), f(act), f(act)
        self.dA2 = f(act) if self.res_mask else None
        self.mean = [f(C) for _ in range(self.n_layers)]
        self.inv = [f(C) for _ in range(self.n_layers)]
        self.scale = [f(C) for _ in range(self.n_layers)]
        self.shift = [f(C) for _ in range(self.n_layers)]
        self.batch_prep = _BATCH_PREP
        self.tail_out = [f(4) for _ in range(self.n_layers)]
        self.relu_mask = [None if self._deferred(l) else f(self.max_rows * self.hw * 32, torch.uint8)
                          for l in range(self.n_layers)]                                                   # 4 bits per byte: out > 0
        self.amax = f(C)
        self.abc = f(3 * C)
        self.img_f = [torch.zeros(F16S_WEIGHT_BYTES, dtype=torch.uint8, device=self.dev) for _ in range(self.n_layers)]
        tail = lambda img: img[F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 16].view(torch.float32)
        if self.batch_prep:
            # one input-gradient image per layer; the scale tails the element-wise kernels write ARE the images' tails
            self.img_b_all = [None] + [torch.zeros(F16S_WEIGHT_BYTES, dtype=torch.uint8, device=self.dev) for _ in range(1, self.n_layers)]
            for l in range(self.n_layers - 1):
                self.tail_out[l] = tail(self.img_f[l + 1])
            self.tail_dy_all = [None] + [tail(self.img_b_all[l]) for l in range(1, self.n_layers)]
            arr = lambda ptrs: (ctypes.c_void_p * len(ptrs))(*ptrs)
            self._h_w = arr([self.view[self._k(l)].data_ptr() for l in range(1, self.n_layers)])
            self._h_f = arr([self.img_f[l].data_ptr() for l in range(1, self.n_layers)])
            self._h_b = arr([self.img_b_all[l].data_ptr() for l in range(1, self.n_layers)])
        else:
            self.img_b_all = [None] + [torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device=self.dev)] * (self.n_layers - 1)
            self.tail_dy_all = [None] + [f(4)] * (self.n_layers - 1)       # (one buffer: the layers take turns)
        self.w_version, self.img_version, self.img_has_bwd = 0, -1, False  # weight images are current when img_version == w_version
        self.ones, self.zeros = torch.ones(C, device=self.dev), f(C)
        self.sums, self.sums_local = f(2 * C, torch.float64), f(2 * C, torch.float64)
        self.partials = f(self.L.snk_bn_train_partials())
        self.partials2 = f(self.L.snk_bn_train_partials())
        self.wg_partials = f(int(self.L.snk_conv3x3_wgrad_partials(self.h, self.w)))
        self.cv_partials = f(int(self.L.snk_conv3x3_stats_partials(self.max_rows, self.h, self.w)))
        self.sw_partials = f(int(self.L.snk_stem_wgrad_partials(self.max_rows, self.h, self.w)))
        # head
        self.z, self.hh, self.g1 = f(self.max_rows * self.hw), f(self.max_rows * self.hw), f(self.max_rows * self.hw)
        self.d1, self.dpre1 = f(self.max_rows * C), f(self.max_rows * C)
        self.q = f(self.max_rows * 3)
        self.h_mean_inv, self.h_sb, self.h_abc = f(2), f(2), f(3)
        self.h_sums, self.h_sums_local = f(2, torch.float64), f(2, torch.float64)
        self.small = f(515)
        self.hb_partials = f(self.L.snk_head_dense_train_bwd_partials(self.max_rows))
        self.l2_partials = f(1024)
        self.l2_value = f(1)
        self.l2_version = -1                                             # the weights' version l2_value was computed for
        self.mask_override = {}          # tests: layer -> tensor whose sign replaces out > 0 as the ReLU mask ('h', 'd1': the head's)
        self.saved_rows = 0

    # ---- small helpers ---------------------------------------------------------------------------------------------
    def _prepare_images(self, want_bwd):
        """the tower's weight images, once per set of weights (the batch form; the other form makes them layer by layer in place)"""
        if not self.batch_prep or (self.img_version == self.w_version and (self.img_has_bwd or not want_bwd)):
            return
        check(self.L.snk_conv3x3_prepare_weights_f16s_train_batch(self._h_w, self._h_f, self._h_b if want_bwd else None,
                                                                  self.n_layers - 1, self._st()))
        self.img_version, self.img_has_bwd = self.w_version, bool(want_bwd)

    def _st(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _deferred(self, l):
        return self.defer and (l % 2 == 1 or (l == 0 and self.defer_stem))

    def _k(self, l):                     # Keras list index of conv layer l's kernel (gamma, beta, moving mean / variance follow)
        return 5 * l

    def _all_reduce(self, t):
        if self.dist is not None:
            if self.dist.get_backend() == "gloo":
                c = t.cpu()
                self.dist.all_reduce(c)
                t.copy_(c)
            else:
                self.dist.all_reduce(t)

    def _bn_forward(self, l, n, count, res, relu_out, tail, have_sums=False, head=False):
        L, st, rows, k = self.L, self._st(), n * self.hw, self._k(l)
        mm, mv = self.moving[k + 3], self.moving[k + 4]
        if not have_sums:                                        # the tower's convolutions take the sums on their way out
            check(L.snk_bn_train_sums_f64(_p(self.y[l]), rows, _p(mm), _p(self.partials), _p(self.sums), st))
        self._all_reduce(self.sums)
        if self._deferred(l):                                    # nobody writes out_l: its range comes from the convolution's maxima
            check(L.snk_bn_train_finalize_range(_p(self.sums), float(count), _p(mm), _p(self.view[k + 1]), _p(self.view[k + 2]), _p(mm),
                                                _p(mv), BN_MOMENTUM, BN_EPS, _p(self.mean[l]), _p(self.inv[l]), _p(self.scale[l]),
                                                _p(self.shift[l]), _p(self.amax), _p(tail), C, st))
            return
        check(L.snk_bn_train_finalize(_p(self.sums), float(count), _p(mm), _p(self.view[k + 1]), _p(self.view[k + 2]), _p(mm), _p(mv),
                                      BN_MOMENTUM, BN_EPS, _p(self.mean[l]), _p(self.inv[l]), _p(self.scale[l]), _p(self.shift[l]), C, st))
        if head and not isinstance(res, tuple):                  # the last layer: the head's 1x1 convolution and its sums ride along
            kh = self._k(self.n_layers)
            check(L.snk_bn_train_apply_head(_p(self.y[l]), _p(self.scale[l]), _p(self.shift[l]), _p(res), _p(relu_out), rows, _p(self.partials),
                                            _p(self.relu_mask[l]), _p(self.view[kh]), _p(self.moving[kh + 3]), _p(self.z), _p(self.h_sums), st))
            return
        if isinstance(res, tuple):                               # the shortcut is a deferred activation: (its y, its scale, its shift)
            check(L.snk_bn_train_apply_res_deferred(_p(self.y[l]), _p(self.scale[l]), _p(self.shift[l]), _p(res[0]), _p(res[1]), _p(res[2]),
                                                    _p(relu_out), rows, _p(self.partials), _p(tail), _p(self.relu_mask[l]), st))
            if head:                                             # (a one-block net: the last layer's shortcut is the deferred stem)
                kh = self._k(self.n_layers)
                check(L.snk_head_conv1x1_sums(_p(relu_out), _p(self.view[kh]), rows, _p(self.moving[kh + 3]), _p(self.z), _p(self.partials),
                                              _p(self.h_sums), st))
            return
        check(L.snk_bn_train_apply(_p(self.y[l]), _p(self.scale[l]), _p(self.shift[l]), _p(res), _p(relu_out), rows, 1,
                                   _p(self.partials), _p(tail), _p(self.relu_mask[l]), st))

    def _conv(self, x, image, res, out, n):
        check(self.L.snk_conv3x3_bn_f16s(_p(x), _p(image), _p(self.ones), _p(self.zeros), _p(res), _p(out), n, self.h, self.w, 0, self._st()))

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward(self, x, target, n_global, want_bwd=False):
        """x: cuda float32 [n, h, w, 3] contiguous, target: [n, 3] or None; n_global: rows of the batch over all ranks.
        Leaves q in self.q[:n * 3]; the squared-error term of the loss (divided by 3 n_global) goes to G[-1]."""
        L, st = self.L, self._st()
        self._prepare_images(want_bwd)
        n = int(x.shape[0])
        assert 0 < n <= self.max_rows and x.is_contiguous() and x.dtype == torch.float32 and tuple(x.shape[1:]) == (self.h, self.w, 3)
        count = n_global * self.hw
        self.x0, self.saved_rows = x, n
        if self._deferred(0):                                     # ... and the maxima its deferred batch norm is ranged by
            check(L.snk_stem_conv_f32_stats_deferred(_p(x), _p(self.view[0]), _p(self.y[0]), _p(self.moving[3]), _p(self.amax), _p(self.partials),
                                                     _p(self.sums), n, self.h, self.w, st))
        elif _CONV_STATS:                                         # the stem's batch-norm sums leave its kernel with the output
            check(L.snk_stem_conv_f32_stats(_p(x), _p(self.view[0]), _p(self.y[0]), _p(self.moving[3]), _p(self.partials), _p(self.sums),
                                            n, self.h, self.w, st))
        else:
            check(L.snk_stem_conv_f32(_p(x), _p(self.view[0]), _p(self.y[0]), n, self.h, self.w, st))
        self._bn_forward(0, n, count, None, self.out[0], self.tail_out[0], have_sums=_CONV_STATS)
        for l in range(1, self.n_layers):
            k = self._k(l)
            if not self.batch_prep:
                check(L.snk_conv3x3_prepare_weights_f16s_train(_p(self.view[k]), _p(self.img_f[l]), _p(self.tail_out[l - 1]), 0, None, st))
            if self._deferred(l - 1):                             # reads y_{l-1} through layer l - 1's batch norm + ReLU
                check(L.snk_conv3x3_f16s_stats_deferred(_p(self.y[l - 1]), _p(self.img_f[l]), _p(self.y[l]), _p(self.moving[k + 3]),
                                                        _p(self.scale[l - 1]), _p(self.shift[l - 1]),
                                                        _p(self.amax) if self._deferred(l) else None, _p(self.cv_partials),
                                                        _p(self.sums), n, self.h, self.w, st))
            elif self._deferred(l):                               # also takes the maxima its own deferred batch norm is ranged by
                check(L.snk_conv3x3_f16s_stats_deferred(_p(self.out[l - 1]), _p(self.img_f[l]), _p(self.y[l]), _p(self.moving[k + 3]),
                                                        None, None, _p(self.amax), _p(self.cv_partials), _p(self.sums), n, self.h,
                                                        self.w, st))
            elif _CONV_STATS:
                check(L.snk_conv3x3_f16s_stats(_p(self.out[l - 1]), _p(self.img_f[l]), _p(self.y[l]), _p(self.moving[k + 3]),
                                               _p(self.cv_partials), _p(self.sums), n, self.h, self.w, st))
            else:
                self._conv(self.out[l - 1], self.img_f[l], None, self.y[l], n)
            res = self.out[l - 2] if l % 2 == 0 else None                # a block's second layer adds the block's input
            if l % 2 == 0 and self._deferred(l - 2):                     # ... which, above a deferred stem, is read from the stem's y
                res = (self.y[l - 2], self.scale[l - 2], self.shift[l - 2])
            self._bn_forward(l, n, count, res, self.out[l], self.tail_out[l], have_sums=_CONV_STATS,
                             head=_HEAD_FUSED and l == self.n_layers - 1)
        kh = self._k(self.n_layers)
        mm, mv = self.moving[kh + 3], self.moving[kh + 4]
        rows = n * self.hw
        if not _HEAD_FUSED:
            check(L.snk_head_conv1x1_sums(_p(self.out[-1]), _p(self.view[kh]), rows, _p(mm), _p(self.z), _p(self.partials), _p(self.h_sums), st))
        self._all_reduce(self.h_sums)
        check(L.snk_bn_train_finalize(_p(self.h_sums), float(count), _p(mm), _p(self.view[kh + 1]), _p(self.view[kh + 2]), _p(mm), _p(mv),
                                      BN_MOMENTUM, BN_EPS, _p(self.h_mean_inv), _p(self.h_mean_inv) + 4, _p(self.h_sb), _p(self.h_sb) + 4, 1, st))
        i = self.fc_idx
        check(L.snk_head_dense_train_fwd(_p(self.z), _p(self.h_sb), _p(self.view[i]), _p(self.view[i + 1]), _p(self.view[i + 2]),
                                         _p(self.view[i + 3]), _p(target), _p(self.hh), _p(self.d1), _p(self.q), _p(self.partials),
                                         _p(self.G) + 4 * self.n_params if target is not None else None, 1.0 / (3.0 * n_global),
                                         n, self.h, self.w, st))
        return self.q[:3 * n].view(n, 3)

    # ---- backward --------------------------------------------------------------------------------------------------
    def _bn_backward(self, l, n, count, want_res, tail, have_sums=False, src=None):
        L, st, rows, k = self.L, self._st(), n * self.hw, self._k(l)
        src = self.dA if src is None else src                     # the gradient at out_l
        sign = self.mask_override.get(l)                          # tests: a tensor whose sign replaces the recorded ReLU mask
        bits = None if sign is not None else self.relu_mask[l]
        deferred = sign is None and self._deferred(l)             # no mask bytes: the decision is recomputed from y_l, scale_l, shift_l
        if not have_sums and deferred:
            check(L.snk_bn_train_grad_sums_f64_deferred(_p(src), _p(self.y[l]), _p(self.scale[l]), _p(self.shift[l]), _p(self.mean[l]),
                                                        _p(self.inv[l]), rows, _p(self.partials), _p(self.sums), st))
        elif not have_sums:                                       # the input-gradient convolution above took them on its way out
            check(L.snk_bn_train_grad_sums_f64(_p(src), _p(sign), _p(bits), _p(self.y[l]), _p(self.mean[l]), _p(self.inv[l]), rows, 1,
                                               _p(self.partials), _p(self.sums), st))
        local = self.sums
        if self.dist is not None:
            self.sums_local.copy_(self.sums)
            local = self.sums_local
            self._all_reduce(self.sums)
        a, b, c = self.abc[:C], self.abc[C:2 * C], self.abc[2 * C:]
        check(L.snk_bn_train_grad_finalize(_p(self.sums), _p(local), float(count), _p(self.view[k + 1]), _p(self.inv[l]), _p(a), _p(b), _p(c),
                                           _p(self.gview[k + 1]), _p(self.gview[k + 2]), C, st))
        if deferred:
            check(L.snk_bn_train_grad_apply_deferred(_p(src), _p(self.y[l]), _p(self.scale[l]), _p(self.shift[l]), _p(self.mean[l]),
                                                     _p(self.inv[l]), _p(a), _p(b), _p(c), _p(self.dY),
                                                     _p(self.gres) if want_res else None, rows, _p(self.partials), _p(tail), st))
            return
        check(L.snk_bn_train_grad_apply(_p(src), _p(sign), _p(bits), _p(self.y[l]), _p(self.mean[l]), _p(self.inv[l]), _p(a), _p(b), _p(c),
                                        _p(self.dY), _p(self.gres) if want_res else None, rows, 1, _p(self.partials), _p(tail), st))

    def backward(self, target, n_global):
        L, st = self.L, self._st()
        self._prepare_images(True)
        n = self.saved_rows
        rows, count = n * self.hw, n_global * self.hw
        i, kh = self.fc_idx, self._k(self.n_layers)
        check(L.snk_head_dense_train_bwd(_p(self.q), _p(target), _p(self.hh), _p(self.d1), _p(self.mask_override.get("h")),
                                         _p(self.mask_override.get("d1")), _p(self.z), _p(self.h_mean_inv), _p(self.view[i]),
                                         _p(self.view[i + 2]), 1.0 / (3.0 * n_global), _p(self.dpre1), _p(self.g1), _p(self.gview[i]),
                                         _p(self.small), _p(self.h_sums), _p(self.hb_partials), n, self.h, self.w, st))
        self.gview[i + 2].copy_(self.small[:384])
        self.gview[i + 3].copy_(self.small[384:387])
        self.gview[i + 1].copy_(self.small[387:515])
        local = self.h_sums
        if self.dist is not None:
            self.h_sums_local.copy_(self.h_sums)
            local = self.h_sums_local
            self._all_reduce(self.h_sums)
        check(L.snk_bn_train_grad_finalize(_p(self.h_sums), _p(local), float(count), _p(self.view[kh + 1]), _p(self.h_mean_inv) + 4,
                                           _p(self.h_abc), _p(self.h_abc) + 4, _p(self.h_abc) + 8, _p(self.gview[kh + 1]),
                                           _p(self.gview[kh + 2]), 1, st))
        top = self.n_layers - 1
        have_sums = _HEAD_FUSED and top not in self.mask_override   # the top layer's backward sums leave with the gradient the head writes
        if have_sums:
            check(L.snk_head_conv1x1_bwd_stats(_p(self.g1), _p(self.z), _p(self.h_mean_inv), _p(self.h_abc), _p(self.out[-1]), _p(self.view[kh]),
                                               _p(self.dA), _p(self.gview[kh]), _p(self.partials), _p(self.y[top]), _p(self.relu_mask[top]),
                                               _p(self.mean[top]), _p(self.inv[top]), _p(self.partials2), _p(self.sums), rows, st))
        else:
            check(L.snk_head_conv1x1_bwd(_p(self.g1), _p(self.z), _p(self.h_mean_inv), _p(self.h_abc), _p(self.out[-1]), _p(self.view[kh]),
                                         _p(self.dA), _p(self.gview[kh]), _p(self.partials), rows, st))
        A, B, masked = self.dA, self.dA2, False                   # A holds the gradient a block receives at its output
        for l in range(self.n_layers - 1, 0, -1):
            k, second = self._k(l), l % 2 == 0
            self.img_b, self.tail_dy = self.img_b_all[l], self.tail_dy_all[l]
            if second:                                            # this block leaves its shortcut's gradient in A, unmasked and unwritten?
                masked = self.res_mask and not {l, l - 2} & set(self.mask_override)
            src = B if (masked and not second) else A
            self._bn_backward(l, n, count, want_res=second and not masked, tail=self.tail_dy, have_sums=have_sums, src=src)   # -> dY (+ gres)
            if self._deferred(l - 1):
                check(L.snk_conv3x3_wgrad_f16s_deferred(_p(self.y[l - 1]), _p(self.scale[l - 1]), _p(self.shift[l - 1]), _p(self.dY),
                                                        _p(self.tail_out[l - 1]), _p(self.tail_dy), _p(self.wg_partials),
                                                        _p(self.gview[k]), n, self.h, self.w, st))
            else:
                check(L.snk_conv3x3_wgrad_f16s(_p(self.out[l - 1]), _p(self.dY), _p(self.tail_out[l - 1]), _p(self.tail_dy),
                                               _p(self.wg_partials), _p(self.gview[k]), n, self.h, self.w, st))
            if not self.batch_prep:
                check(L.snk_conv3x3_prepare_weights_f16s_train(_p(self.view[k]), _p(self.img_b), _p(self.tail_dy), 1, _p(self.img_f[l]), st))
            # gradient at out[l - 1]; its epilogue also takes the two sums the batch-norm backward of layer l - 1 starts with
            have_sums = _IGRAD_STATS and (l - 1) not in self.mask_override
            dst = B if (masked and second) else A                 # (a block's second layer must not overwrite what its shortcut still needs)
            res = None if second else (A if masked else self.gres)
            if have_sums and not second and not masked and self._deferred(l - 1):
                have_sums = False                                 # (deferred stem below, shortcut as a masked copy: no launch takes both)
            if masked and not second and self._deferred(l - 1):
                check(L.snk_conv3x3_f16s_igrad_stats_masked_res_deferred(_p(self.dY), _p(self.img_b), _p(A), _p(self.relu_mask[l + 1]), _p(A),
                                                                         _p(self.y[l - 1]), _p(self.scale[l - 1]), _p(self.shift[l - 1]),
                                                                         _p(self.mean[l - 1]), _p(self.inv[l - 1]), _p(self.cv_partials),
                                                                         _p(self.sums), n, self.h, self.w, st))
            elif masked and not second:
                check(L.snk_conv3x3_f16s_igrad_stats_masked_res(_p(self.dY), _p(self.img_b), _p(A), _p(self.relu_mask[l + 1]), _p(A),
                                                                _p(self.y[l - 1]), _p(self.relu_mask[l - 1]), _p(self.mean[l - 1]),
                                                                _p(self.inv[l - 1]), _p(self.cv_partials), _p(self.sums), n, self.h,
                                                                self.w, st))
            elif have_sums and self._deferred(l - 1):
                check(L.snk_conv3x3_f16s_igrad_stats_deferred(_p(self.dY), _p(self.img_b), _p(res), _p(dst),
                                                              _p(self.y[l - 1]), _p(self.scale[l - 1]), _p(self.shift[l - 1]),
                                                              _p(self.mean[l - 1]), _p(self.inv[l - 1]), _p(self.cv_partials),
                                                              _p(self.sums), n, self.h, self.w, st))
            elif have_sums:
                check(L.snk_conv3x3_f16s_igrad_stats(_p(self.dY), _p(self.img_b), _p(res), _p(dst),
                                                     _p(self.y[l - 1]), _p(self.relu_mask[l - 1]), _p(self.mean[l - 1]), _p(self.inv[l - 1]),
                                                     _p(self.cv_partials), _p(self.sums), n, self.h, self.w, st))
            else:
                self._conv(self.dY, self.img_b, res, dst, n)
        self._bn_backward(0, n, count, want_res=False, tail=None, have_sums=have_sums)
        check(L.snk_stem_wgrad_f32(_p(self.x0), _p(self.dY), _p(self.sw_partials), _p(self.gview[0]), n, self.h, self.w, st))

    # ---- the two kinds of step ----------------------------------------------------------------------------------------
    def step(self, x, target, n_global, lr):
        """forward, backward, gradient all-reduce, Adam.  Returns a 2-element device tensor {mse, l2 loss} of this step."""
        L, st = self.L, self._st()
        self.forward(x, target, n_global, want_bwd=True)
        self.backward(target, n_global)
        self._all_reduce(self.G)                                       # every gradient + the squared-error term, one bucket
        check(L.snk_l2_sum(_p(self.W), _p(self.decay), self.n_params, L2_C, _p(self.l2_partials), _p(self.l2_value), st))
        self.l2_version = self.w_version
        loss = torch.stack([self.G[self.n_params], self.l2_value[0]])
        self.adam_t += 1
        lr_t = lr * np.sqrt(1.0 - ADAM_B2 ** self.adam_t) / (1.0 - ADAM_B1 ** self.adam_t)
        check(L.snk_adam_l2_step(_p(self.W), _p(self.G), _p(self.M), _p(self.V), _p(self.decay), self.n_params, float(lr_t),
                                 ADAM_B1, ADAM_B2, ADAM_EPS, L2_C, st))
        self.w_version += 1                                            # the weight images are stale
        return loss

    def forward_only(self, x, target, n_global):
        """a step at learning rate 0: the weights cannot move, the batch-norm moving averages do"""
        L, st = self.L, self._st()
        self.forward(x, target, n_global)
        mse = self.G[self.n_params:self.n_params + 1]
        self._all_reduce(mse)
        if self.l2_version != self.w_version:                          # (the weights of a forward-only step are those of the step before)
            check(L.snk_l2_sum(_p(self.W), _p(self.decay), self.n_params, L2_C, _p(self.l2_partials), _p(self.l2_value), st))
            self.l2_version = self.w_version
        self.adam_t += 1
        return torch.stack([mse[0], self.l2_value[0]])

    def gradients(self):
        """{Keras list index: gradient} of the last `backward` (without the regularizer's term), host arrays"""
        return {j: self.gview[j].cpu().numpy().reshape(self.shape[j]).copy() for j in self.param_idx}

    def weights(self):
        """the Keras-order weight list, host arrays"""
        n = len(self.view) + len(self.moving)
        out = []
        for j in range(n):
            if j in self.view:
                out.append(self.view[j].cpu().numpy().reshape(self.shape[j]).copy())
            else:
                out.append(self.moving[j].cpu().numpy().copy())
        return out
