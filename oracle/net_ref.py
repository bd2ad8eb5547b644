"""Plain PyTorch fp32 CPU restatement of the reference Q-net (TEST INFRASTRUCTURE).

Restates AlphaNNet.__init__'s Keras graph (alpha_nnet.py:19-56) and AlphaNNet.v (alpha_nnet.py:61-76).
PARITY UNPINNED at this boundary: the reference's arithmetic lives in TensorFlow/Keras 2.1 (README.md:16-20,
no lockfile), which is absent from this image and cannot be installed offline, and the reference commits no
.h5 file or net output to pin against.  This module follows the published Keras layer definitions
(Conv2D 'same' no bias = cross-correlation, BatchNormalization inference gamma*(x-mean)/sqrt(var+1e-3)+beta,
Dense = x @ kernel + bias) and is the oracle for "Q within 1e-5".
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def forward(weights, states, legacy_mask=False, apply_mask=True, bf16_conv=False, f16_act=False, bf16_act=False):
    """weights: list in Keras get_weights() order (snake_engine/net.py docstring); states (N,h,w,3) float32.
    Returns (N,3) float32 numpy: AlphaNNet.v(states)."""
    x_np = np.ascontiguousarray(states, np.float32)
    t = [torch.as_tensor(np.asarray(w, np.float32)) for w in weights]
    blocks = (len(t) - 14) // 10
    x = torch.as_tensor(x_np).permute(0, 3, 1, 2)                      # NHWC -> NCHW

    def conv(x, k, bf16=False):                                         # Keras (kh,kw,cin,cout) -> torch (cout,cin,kh,kw)
        if bf16:        # configs[4]: operands rounded to bf16 (round to nearest even), products and sums in float32
            x, k = x.to(torch.bfloat16).to(torch.float32), k.to(torch.bfloat16).to(torch.float32)
        return F.conv2d(x, k.permute(3, 2, 0, 1), padding=k.shape[0] // 2)

    def bn(x, g, b, m, v):
        return F.batch_norm(x, m, v, g, b, training=False, eps=BN_EPS)
    def r16(v):      # f16_act / bf16_act: the tower's activations live in HBM as f16 / bf16 (round to nearest even), the last layer's output stays float32
        if bf16_act:
            return v.to(torch.bfloat16).to(torch.float32)
        return v.to(torch.float16).to(torch.float32) if f16_act else v

    def conv16(x, k):   # ... and the tower convolutions take f16-rounded weights (scaled by the power of two that brings max|w| to [256, 512))
        if bf16_act:    # bf16 weights (a power-of-two scale does not move a bf16 rounding), the activations are bf16 numbers already
            return conv(x, k.to(torch.bfloat16).to(torch.float32))
        if not f16_act:
            return conv(x, k, bf16_conv)
        sc_ = 2.0 ** (8 - int(np.floor(np.log2(float(k.abs().max())))))
        return conv(x, (k * sc_).to(torch.float16).to(torch.float32) / sc_)
    with torch.no_grad():
        h = r16(F.relu(bn(conv(x, t[0]), *t[1:5])))
        for blk in range(blocks):
            b0 = 5 + 10 * blk
            sc = h
            h = r16(F.relu(bn(conv16(h, t[b0]), *t[b0 + 1:b0 + 5])))
            h = F.relu(bn(conv16(h, t[b0 + 5]), *t[b0 + 6:b0 + 10]) + sc)
            if blk + 1 < blocks:
                h = r16(h)
        b0 = 5 + 10 * blocks
        h = F.relu(bn(conv(h, t[b0]), *t[b0 + 1:b0 + 5]))              # (N,1,h,w)
        h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)              # Flatten in HWC order
        h = F.relu(h @ t[b0 + 5] + t[b0 + 6])
        q = torch.tanh(h @ t[b0 + 7] + t[b0 + 8]).numpy()
    if apply_mask:                                                     # alpha_nnet.py:63-76
        from oracle.obs_key import obstacle_mask
        q[obstacle_mask(x_np, legacy_mask)] = np.float32(-1.0)
    return q
