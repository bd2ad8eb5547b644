"""QNet: the reference's Q-network (alpha_nnet.py:19-56) as a list of Keras-layout weight arrays
plus a forward pass made of the hand-written HIP kernels of csrc/net.hip.

Weight list order (= Keras ``model.get_weights()`` order of the reference graph):
  stem kernel (3,3,3,k), stem BN gamma/beta/mean/var,
  for each of the 2*blocks residual convs: kernel (3,3,k,k), BN gamma/beta/mean/var,
  head kernel (1,1,k,1), head BN (4 x (1,)), dense kernel (h*w, 128), bias, dense_1 kernel (128, 3), bias.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from ._lib import lib, check, EngineError

BN_EPS = 1e-3        # Keras BatchNormalization default epsilon (alpha_nnet.py:22)
K_FILTERS = 128      # alpha_nnet.py:17
F16S_WEIGHT_BYTES = 9 * 128 * 128 * 4 + 32     # SNK_CONV_F16S_WEIGHT_BYTES (include/snake_engine.h)
F16S_TAIL_OFFSET = 9 * 128 * 128 * 4           # float32 {2^-k, 2^k, x_scale, 1/x_scale}, int32 range flag, padding
F16S_FLAG_OFFSET = F16S_TAIL_OFFSET + 16


def glorot_uniform_weights(input_shape, blocks=4, seed=0):
    """Keras defaults: glorot_uniform kernels, zero biases, BN gamma 1 / beta 0 / mean 0 / var 1."""
    g = torch.Generator().manual_seed(seed)
    h, w, c = input_shape
    k = K_FILTERS

    def glorot(shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return ((torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * lim).numpy()

    def bn(n):
        return [np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32), np.ones(n, np.float32)]
    ws = [glorot((3, 3, c, k), 9 * c, 9 * k)] + bn(k)
    for _ in range(2 * blocks):
        ws += [glorot((3, 3, k, k), 9 * k, 9 * k)] + bn(k)
    ws += [glorot((1, 1, k, 1), k, 1)] + bn(1)
    ws += [glorot((h * w, 128), h * w, 128), np.zeros(128, np.float32)]
    ws += [glorot((128, 3), 128, 3), np.zeros(3, np.float32)]
    return ws


def activation_scales(weights, target_log2=9):
    """Power-of-two scale of every tower layer's INPUT for the split-f16 kernel (snk_conv3x3_prepare_weights_f16s).
    A bound on the input comes from the producing batch-norm: |beta| + 8 |gamma| (eight standard deviations of a
    normalised pre-activation), plus the shortcut's own bound behind a residual add.  The scale brings the bound to at
    most 2^target_log2: 2^9 leaves 128x headroom to the f16 maximum (beyond it the kernel clamps) and keeps the lo
    part a normal f16 number for every input above 1/2048 of the bound."""
    blocks = n_blocks_of(weights)

    def bn_bound(g, b):
        return float(np.max(np.abs(np.asarray(b, np.float64)) + 8.0 * np.abs(np.asarray(g, np.float64))))
    bound_in = bn_bound(weights[1], weights[2])              # stem output
    scales = []
    bound_block_in = bound_in
    for i in range(2 * blocks):
        base = 5 + 5 * i
        scales.append(2.0 ** (target_log2 - math.ceil(math.log2(max(bound_in, 2.0 ** -100)))))
        if i % 2 == 0:
            bound_block_in, bound_in = bound_in, bn_bound(weights[base + 1], weights[base + 2])
        else:
            bound_in = bound_block_in + bn_bound(weights[base + 1], weights[base + 2])       # relu(bn(conv) + shortcut)
    return scales


def rect_fill_plan(n_rect):
    """Which sub-rectangle layers also write their background into the rest of the canvas.  The output of tower layer i is
    read by layer i + 1 and, when it closes a residual block (odd i), by layer i + 2 as the shortcut.  A sub-rectangle reader
    takes what lies outside its producer's rectangle from the producer's background image; a FULL reader reads the tensor
    everywhere, so a sub-rectangle layer with a full reader fills the canvas."""
    return [i + 1 >= n_rect or (i % 2 == 1 and i + 2 >= n_rect) for i in range(n_rect)]


class ClockProbe:
    """Samples of the shader clock while other streams work (csrc/probe.hip): every `launch` puts one wavefront on a
    stream of its own that sits on a compute unit for `microseconds` and records shader cycles against the constant
    100 MHz counter.  `mhz()` -> the samples taken so far."""

    def __init__(self, device, capacity=8192, microseconds=200):
        self.L = lib()
        self.buf = torch.zeros((capacity, 2), dtype=torch.int64, device=device)
        self.stream = torch.cuda.Stream(device=device)
        self.us, self.n = int(microseconds), 0

    def launch(self):
        if self.n < self.buf.shape[0]:
            check(self.L.snk_clock_probe(self.buf[self.n].data_ptr(), self.us, self.stream.cuda_stream))
            self.n += 1

    def mhz(self):
        self.stream.synchronize()
        a = self.buf[:self.n].cpu().numpy().astype(np.float64)
        a = a[a[:, 1] > 0]
        return a[:, 0] / a[:, 1] * 100.0


def n_blocks_of(weights):
    return (len(weights) - 14) // 10


BACKGROUND_PIXEL = (0.0, 1.0, 0.0)      # game.py:218: every pixel of an observation outside the board window is [0, WALL, 0]


def rect_layer_count(h, w, n_layers, threshold=0.93):
    """How many of the tower's first layers run in the sub-rectangle form by default: tower layer i needs the board window
    (B x B, B = (h + 1) / 2, somewhere on the canvas with the head in the centre) grown by i + 2 pixels; the form pays while
    the grown window, averaged over the head positions, stays below `threshold` of the canvas.  The last layer always
    runs in full (its epilogue feeds the head)."""
    bh, bw = (h + 1) // 2, (w + 1) // 2
    count = 0
    for i in range(max(n_layers - 1, 0)):
        g = i + 2
        ly = [min(h - 1, s + bh - 1 + g) - max(0, s - g) + 1 for s in range(h - bh + 1)]
        lx = [min(w - 1, s + bw - 1 + g) - max(0, s - g) + 1 for s in range(w - bw + 1)]
        if (sum(ly) / len(ly)) * (sum(lx) / len(lx)) >= threshold * h * w:
            break
        count += 1
    return count


class QNet:
    def __init__(self, weights, input_shape, device=None, max_chunk=4096):
        if not torch.cuda.is_available():
            raise EngineError("snake_engine.QNet needs an MI355X; there is no CPU fallback")
        self.L = lib()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.input_shape = tuple(int(v) for v in input_shape)
        self.h, self.w, self.cin = self.input_shape
        assert self.cin == 3
        self.max_chunk = int(max_chunk)
        self._ws = None
        self.conv_timing = None      # set to a list to collect (start_event, end_event, flops) per conv3x3 launch
        self.clock_probe = None      # set to a ClockProbe to sample the chip's clock beside every forward chunk
        # "f16s" (default): float32-accurate split-f16 MFMA kernel (csrc/conv_split.hip); "winograd": F(2x2,3x3) fp32 MFMA
        # kernel; "direct": implicit-GEMM fp32 MFMA kernel; reduced precision for configs[4], NOT within the 1e-5 parity
        # tolerance: "f16" (the f16s kernel with the hi parts only, one MFMA per product, float32 activations)
        self.conv_algo = os.environ.get("SNK_CONV_ALGO", "f16s")
        # "f16a": "f16" with the tower's activations stored as f16 in HBM (half the traffic; the f16 form with float32
        # activations is HBM-bound); "bf16": the same block body on bf16 -- bf16 activations in HBM, bf16 weights,
        # v_mfma_f32_32x32x16_bf16: BASELINE configs[4]'s "bf16 MFMA conv" as it is worded
        if self.conv_algo not in ("f16s", "winograd", "direct", "bf16", "f16", "f16a"):
            raise EngineError(f"SNK_CONV_ALGO={self.conv_algo!r}: expected f16s, winograd, direct, f16, f16a or bf16")
        # sub-rectangle form of the first tower layers (snk_conv3x3_bn_f16s_rect): SNK_CONV_RECT=0 switches it off,
        # SNK_CONV_RECT_LAYERS=n fixes the number of layers that use it
        self.rect = self.conv_algo in ("f16s", "f16a", "bf16") and os.environ.get("SNK_CONV_RECT", "1") != "0"
        self.act16 = {"f16a": torch.float16, "bf16": torch.bfloat16}.get(self.conv_algo)      # 16-bit activations in HBM
        self.background = BACKGROUND_PIXEL
        self.guard_trips = 0         # batches forward_guarded evaluated again after a clamp
        self.rect_tiles = None       # set to [] to collect every chunk's (images, per-layer GEMM tiles) device tensors
        self._bg = None
        # chunks below this many observations take the full form: a launch that small is bound by one block's duration (the
        # full form then cuts the images into one-tile blocks, conv_split.hip `fine_max`), and the plan's two launches
        # cost as much as a layer (BASELINE configs[0]: 8 games)
        self.rect_min = int(os.environ.get("SNK_CONV_RECT_MIN", "48"))
        self.n_streams = int(os.environ.get("SNK_NET_STREAMS", "1"))   # 2: chunks alternate between two streams (+0.8 % end to end,
        #    but per-launch HIP-event timings then overlap, so bench.py keeps the single-stream default)
        self._side = None
        self._guard = None
        self.set_weights(weights)

    # ---- weights -----------------------------------------------------------------------------
    def set_weights(self, weights):
        self.weights = [np.ascontiguousarray(w, np.float32) for w in weights]
        self.blocks = n_blocks_of(self.weights)
        dev = self.device
        t = [torch.as_tensor(w, device=dev) for w in self.weights]

        def fold(g, b, m, v):
            sc = g / torch.sqrt(v + BN_EPS)
            return sc.contiguous(), (b - m * sc).contiguous()
        self.stem_w = t[0].contiguous()
        self.stem_sc, self.stem_sh = fold(*t[1:5])
        self.conv_wT, self.conv_sc, self.conv_sh = [], [], []
        st = torch.cuda.current_stream().cuda_stream
        self.conv_x_scale = activation_scales(self.weights)      # used by the "f16s" kernel only
        # the split-f16 weight images of all layers are rows of ONE buffer, so that the layers' range flags (a word in each
        # image's tail) come to the host with one strided copy (range_flags)
        self._wimg = None
        if self.conv_algo in ("f16s", "f16", "f16a", "bf16") and self.blocks:
            self._wimg = torch.empty((2 * self.blocks, F16S_WEIGHT_BYTES), dtype=torch.uint8, device=dev)
            self._flags_host = torch.empty((2 * self.blocks,), dtype=torch.int32).pin_memory()
            if self._guard is None:      # ONE device word every layer of this net reports a clamp to, and its pinned host mirror
                self._guard = (torch.zeros((1,), dtype=torch.int32, device=dev), torch.zeros((1,), dtype=torch.int32).pin_memory())
        for i in range(2 * self.blocks):
            base = 5 + 5 * i
            if self.conv_algo == "bf16":
                self.conv_x_scale[i] = 1.0
                wT = self._wimg[i]
                check(self.L.snk_conv3x3_prepare_weights_bf16(t[base].contiguous().data_ptr(), wT.data_ptr(), st))
            elif self.conv_algo == "f16a":
                self.conv_x_scale[i] = 1.0                   # f16 activations are staged as they are
                wT = self._wimg[i]
                check(self.L.snk_conv3x3_prepare_weights_f16_act16(t[base].contiguous().data_ptr(), wT.data_ptr(), st))
            elif self.conv_algo in ("f16s", "f16"):
                wT = self._wimg[i]
                check(self.L.snk_conv3x3_prepare_weights_f16s(t[base].contiguous().data_ptr(), wT.data_ptr(),
                                                              self.conv_x_scale[i], st))
            elif self.conv_algo == "winograd":
                wT = torch.empty(16 * 128 * 128, dtype=torch.float32, device=dev)
                check(self.L.snk_conv3x3_prepare_weights_winograd(t[base].contiguous().data_ptr(), wT.data_ptr(), st))
            else:
                wT = torch.empty(9 * 128 * 128, dtype=torch.float32, device=dev)
                check(self.L.snk_conv3x3_prepare_weights(t[base].contiguous().data_ptr(), wT.data_ptr(), st))
            if self._wimg is not None and self.conv_algo != "bf16":
                check(self.L.snk_conv3x3_f16s_set_guard_word(wT.data_ptr(), self._guard[0].data_ptr(), st))
            sc, sh = fold(*t[base + 1:base + 5])
            self.conv_wT.append(wT); self.conv_sc.append(sc); self.conv_sh.append(sh)
        base = 5 + 10 * self.blocks
        self.head_w = t[base].reshape(128).contiguous()
        hs, hb = fold(*t[base + 1:base + 5])
        self.head_s, self.head_b = float(hs.item()), float(hb.item())
        self.fc1_w, self.fc1_b = t[base + 5].contiguous(), t[base + 6].contiguous()
        self.fc2_w, self.fc2_b = t[base + 7].contiguous(), t[base + 8].contiguous()
        self.calibrated = False                 # new weights: the next caller with observations at hand calibrates
        self._bg = None
        n_layers = 2 * self.blocks
        self.n_rect = rect_layer_count(self.h, self.w, n_layers) if self.rect else 0
        if self.rect and "SNK_CONV_RECT_LAYERS" in os.environ:
            self.n_rect = max(0, min(int(os.environ["SNK_CONV_RECT_LAYERS"]), n_layers - 1))
        self.rect_fill = rect_fill_plan(self.n_rect)
        torch.cuda.current_stream().synchronize()

    def get_weights(self):
        return [w.copy() for w in self.weights]

    # ---- range guard of the split-f16 kernel -------------------------------------------------------
    def _tail(self, i, dtype):
        return self.conv_wT[i][F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 32].view(dtype)

    def set_x_scale(self, i, scale):
        """activation scale of tower layer i (a power of two); only the weight image's tail changes"""
        m, _ = math.frexp(scale)
        assert scale > 0 and m == 0.5, f"x_scale {scale} is not a power of two"
        self.conv_x_scale[i] = float(scale)
        t = self._tail(i, torch.float32)
        t[2] = float(scale)
        t[3] = 1.0 / float(scale)
        self._bg = None                         # the backgrounds are kept bit-identical to what the full layers give

    def range_flags(self, clear=True):
        """per tower layer: 1 when a launch since the last call clamped an input (one small device-to-host copy)"""
        if self._wimg is None:
            return []
        flags = self._wimg[:, F16S_FLAG_OFFSET:F16S_FLAG_OFFSET + 4].contiguous().view(torch.int32).reshape(-1)
        self._flags_host.copy_(flags, non_blocking=True)        # one gather kernel + one copy into pinned memory
        torch.cuda.current_stream().synchronize()
        out = self._flags_host.tolist()
        if clear:
            self._guard[0].zero_()          # the net's guard word says "some flag is set": cleared with them
            self._guard[1].zero_()
            for i, f in enumerate(out):
                if f:
                    self._tail(i, torch.int32)[4] = 0
        return out

    F16A_SATURATED = ("f16-activation tower: outputs of tower layer(s) {} exceeded the f16 range and were saturated; use "
                      "SNK_CONV_ALGO=bf16 (float32's exponent range) or the float32-accurate default")

    def widen(self, layers, factor=2.0 ** -6):
        if layers and self.conv_algo == "f16a":
            # the 16-bit frame never multiplies by the activation scale, but its epilogue multiplies by the tail's inverse: a
            # "lowered" scale would silently scale that layer's outputs by 2^6 from then on.  There is nothing to widen.
            raise EngineError(self.F16A_SATURATED.format(list(layers)))
        for i in layers:
            self.set_x_scale(i, self.conv_x_scale[i] * factor)

    def check_range(self, on_overflow="raise"):
        """Reads the range flags.  Returns the list of layers that clamped (empty = every result since the last check is
        float32-accurate).  Those layers' scales are lowered by 2^6 either way, so a repeated evaluation fits;
        on_overflow="raise" then raises (the results computed in between are not within the 1e-5 contract),
        "widen" leaves the decision to the caller (AlphaNNet.v re-evaluates the batch)."""
        bad = [i for i, f in enumerate(self.range_flags()) if f]
        if bad:
            self.widen(bad)
            if on_overflow == "raise":
                raise EngineError(
                    f"split-f16 convolution: inputs of tower layer(s) {bad} exceeded the f16 range after scaling and were "
                    "clamped, so Q values computed since the last check are not float32-accurate.  The layers' activation "
                    "scales have been lowered (x 2^-6): evaluate again, call QNet.calibrate(observations) after loading "
                    "weights, or use SNK_CONV_ALGO=winograd (no range limit).")
        return bad

    def calibrate(self, planes, target_log2=9):
        """Measures every tower layer's largest |input| on the given observations and lowers the activation scales the
        batch-norm heuristic chose (activation_scales) wherever the measured maximum would come within 2^4 of the f16
        limit: a net whose moving statistics do not describe its activations (freshly trained, loaded from a checkpoint)
        gets scales from data instead.  Returns the activation report."""
        if self.conv_algo not in ("f16s", "f16"):
            self.calibrated = True              # nothing to fit: callers must not come back with observations every turn
            return []
        rep = self.activation_report(planes)
        for i, (amax, scale, _) in enumerate(rep):
            if amax * scale > 2.0 ** 12:
                self.set_x_scale(i, 2.0 ** (target_log2 - math.ceil(math.log2(amax))))
        self.range_flags()                      # the measuring pass itself may have tripped them
        self.calibrated = True
        return rep

    # ---- forward -------------------------------------------------------------------------------
    def _workspace(self, n, k=0):
        if self._ws is None:
            self._ws = {}
        if k not in self._ws or self._ws[k][0].shape[0] < n:
            shape = (n, self.h, self.w, 128)
            self._ws[k] = [torch.empty(shape, dtype=torch.float32, device=self.device) for _ in range(3)]
        return self._ws[k]

    # ---- the range guard as ONE word: csrc/conv_split.hip stores 1 to `guard_ptr` when a launch clamps --------------------------
    @property
    def guard_ptr(self):
        """address of the net's device guard word (0: this net has no range to watch): the gate of the rollout tick's kernels"""
        return self._guard[0].data_ptr() if self._guard is not None and self.conv_algo in ("f16s", "f16", "f16a") else 0

    def guard_post(self):
        """after a forward: copy the guard word to its pinned host mirror, asynchronously -- `guard_tripped` is valid once the
        stream has been synchronised past this point (the search reads it at its next tick's existing read-back)"""
        self._guard[1].copy_(self._guard[0], non_blocking=True)

    def guard_tripped(self):
        return bool(self._guard[1][0])

    def guard_recover(self):
        """a launch clamped: lower the flagged layers' activation scales by 2^6, clear the flags and the word"""
        bad = [i for i, f in enumerate(self.range_flags()) if f]      # also clears the word and its mirror
        # (f16a: f16 activations in HBM have no scale to lower -- an output beyond 65 504 was saturated and nothing can evaluate the
        # batch exactly in this form; widen() says so instead of handing on values computed from saturated activations)
        self.widen(bad)
        self.guard_trips += 1
        return bad

    def forward_guarded(self, planes, mask=None, out=None, tries=4):
        """forward() whose result is float32-accurate or an error, for callers that can wait: after the batch the guard word is
        copied back and the stream synchronised; a layer that clamped an input gets its activation scale lowered by 2^6 and
        the WHOLE batch is evaluated again, so the caller never sees a Q value computed from clamped activations.  (f16s and
        f16: scaled float32 inputs; f16a: an f16 output that saturated raises, there is no scale to lower; winograd, direct and
        bf16 have no range to watch and return forward() as it is.  The search does not wait: its tick kernels are gated on
        the word instead, snake_engine/mcts.py.)"""
        if not self.guard_ptr:                 # no range to watch / a net without tower layers
            return self.forward(planes, mask, out)
        bad = []
        for _ in range(tries):
            out = self.forward(planes, mask, out)
            self.guard_post()
            torch.cuda.current_stream().synchronize()
            if not self.guard_tripped():
                return out
            bad = self.guard_recover()
            if not bad:                      # the word was left over from an unguarded launch whose flags somebody has read since
                return out
        raise EngineError(f"split-f16 convolution: inputs of tower layer(s) {bad} still exceed the f16 range after {tries} "
                          "widenings of their activation scale; use SNK_CONV_ALGO=winograd (no range limit)")

    def forward(self, planes, mask=None, out=None):
        """planes: cuda float32 [n, h, w, 3] (NHWC, contiguous); mask: optional cuda uint8 [n, 3].
        Returns cuda float32 [n, 3] = AlphaNNet.v's output (obstacle entries -1.0 when mask is given).
        Batches larger than max_chunk are cut into chunks; with SNK_NET_STREAMS=2 the chunks alternate between two HIP
        streams (each with its own activation workspace), so the memory-bound stem / head kernels of one chunk run
        under the other chunk's MFMA-bound convolutions."""
        assert planes.is_cuda and planes.dtype == torch.float32 and planes.is_contiguous()
        n = planes.shape[0]
        assert tuple(planes.shape[1:]) == self.input_shape, planes.shape
        if out is None:
            out = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        if self._use_rect(min(n, self.max_chunk)):
            self.backgrounds()       # made on the caller's stream, before any chunk (on whatever stream) reads them
        n_chunks = (n + self.max_chunk - 1) // self.max_chunk
        if n_chunks == 1 or self.n_streams < 2:
            for s0 in range(0, n, self.max_chunk):
                self._forward_chunk(planes, mask, out, s0, min(self.max_chunk, n - s0), 0)
            return out
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        for sd in self._side:
            sd.wait_stream(main)
        for ci, s0 in enumerate(range(0, n, self.max_chunk)):
            with torch.cuda.stream(self._side[ci & 1]):
                self._forward_chunk(planes, mask, out, s0, min(self.max_chunk, n - s0), ci & 1)
        for sd in self._side:
            main.wait_stream(sd)
        return out

    def _fn16(self):
        """(stem, stem on rectangles, layer, layer on rectangles) of the tower with 16-bit activations"""
        L = self.L
        if self.conv_algo == "bf16":
            return (L.snk_stem_conv_bn_relu_bf16out, L.snk_stem_conv_bn_relu_bf16out_rect, L.snk_conv3x3_bn_bf16_act16,
                    L.snk_conv3x3_bn_bf16_act16_rect, L.snk_conv3x3_bn_bf16_act16_head)
        return (L.snk_stem_conv_bn_relu_f16out, L.snk_stem_conv_bn_relu_f16out_rect, L.snk_conv3x3_bn_f16_act16,
                L.snk_conv3x3_bn_f16_act16_rect, L.snk_conv3x3_bn_f16_act16_head)

    def _forward_chunk_f16a(self, planes, mask, out, s0, m, k):
        """the tower with f16 / bf16 activations in HBM: stem -> 16 bit, every layer 16 -> 16 bit, the last one 16 bit -> float32"""
        st = torch.cuda.current_stream().cuda_stream
        L, h, w = self.L, self.h, self.w
        stem16, stem16_rect, conv16, conv16_rect, conv16_head = self._fn16()
        # the last layer's epilogue does the head's 1x1 stage (its float32 output -- twice the bytes of any other activation of
        # this tower -- never goes to HBM); SNK_HEAD_FUSE=0: the separate head kernel on the layer's float32 output
        fused_head = self._head_fits() and os.environ.get("SNK_HEAD_FUSE", "1") != "0"
        key = ("a16", k)
        if self._ws is None:
            self._ws = {}
        if key not in self._ws or self._ws[key][0].shape[0] < m or (self._ws[key][3] is None) != fused_head:
            bufs = [torch.empty((m, h, w, 128), dtype=self.act16, device=self.device) for _ in range(3)]
            bufs.append(None if fused_head else torch.empty((m, h, w, 128), dtype=torch.float32, device=self.device))
            self._ws[key] = bufs
        bufs = self._ws[key]
        a, b, c, last = bufs
        x = planes[s0:s0 + m]
        plan = self._rect_plan(x, m, k, st) if self._use_rect(m) else None
        if plan is not None and self.n_rect >= 2:
            check(stem16_rect(x.data_ptr(), self.stem_w.data_ptr(), self.stem_sc.data_ptr(),
                              self.stem_sh.data_ptr(), a.data_ptr(), plan[3].data_ptr(), 1, m, h, w, st))
        else:
            check(stem16(x.data_ptr(), self.stem_w.data_ptr(), self.stem_sc.data_ptr(),
                         self.stem_sh.data_ptr(), a.data_ptr(), m, h, w, st))
        cur, t1, t2 = a, b, c
        tm = self.conv_timing

        def conv(i, xin, res, dst, out_f16):
            if tm is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
            if dst is None:                                # the last layer with the fused head: dst_h1 instead of a layer output
                check(conv16_head(xin.data_ptr(), self.conv_wT[i].data_ptr(), self.conv_sc[i].data_ptr(), self.conv_sh[i].data_ptr(),
                                  res.data_ptr(), self.head_w.data_ptr(), self.head_s, self.head_b, h1.data_ptr(), m, h, w, st))
            elif plan is not None and i < self.n_rect:     # a sub-rectangle layer is never the last one: f16 output
                check(conv16_rect(xin.data_ptr(), self.conv_wT[i].data_ptr(), self.conv_sc[i].data_ptr(),
                                  self.conv_sh[i].data_ptr(), None if res is None else res.data_ptr(),
                                  dst.data_ptr(), *self._rect_args(i, plan, res), m, h, w, st))
            else:
                check(conv16(xin.data_ptr(), self.conv_wT[i].data_ptr(), self.conv_sc[i].data_ptr(),
                             self.conv_sh[i].data_ptr(), None if res is None else res.data_ptr(),
                             dst.data_ptr(), int(out_f16), m, h, w, 1, st))
            if tm is not None:
                e1.record(torch.cuda.current_stream())
                tm.append((e0, e1, 2.0 * m * h * w * 9 * 128 * 128))
        h1 = self._h1_workspace(m, k) if fused_head else None
        for blk in range(self.blocks):
            conv(2 * blk, cur, None, t1, True)
            if blk == self.blocks - 1:
                conv(2 * blk + 1, t1, cur, last, False)
            else:
                conv(2 * blk + 1, t1, cur, t2, True)
                cur, t2 = t2, cur
        mk = None if mask is None else mask[s0:s0 + m]
        if fused_head:
            check(L.snk_head_dense_f32(h1.data_ptr(), self.fc1_w.data_ptr(), self.fc1_b.data_ptr(), self.fc2_w.data_ptr(),
                                       self.fc2_b.data_ptr(), 0 if mk is None else mk.data_ptr(), out[s0:s0 + m].data_ptr(), m, h, w, st))
            return
        check(L.snk_head_f32(last.data_ptr(), self.head_w.data_ptr(), self.head_s, self.head_b,
                             self.fc1_w.data_ptr(), self.fc1_b.data_ptr(), self.fc2_w.data_ptr(), self.fc2_b.data_ptr(),
                             0 if mk is None else mk.data_ptr(), out[s0:s0 + m].data_ptr(), m, h, w, st))

    def _forward_chunk(self, planes, mask, out, s0, m, k):
        if self.clock_probe is not None:          # bench.py: a one-wavefront clock sample beside this chunk's kernels
            self.clock_probe.launch()
        if self.act16 is not None and self.blocks > 0:
            return self._forward_chunk_f16a(planes, mask, out, s0, m, k)
        st = torch.cuda.current_stream().cuda_stream
        L, h, w = self.L, self.h, self.w
        a, b, c = self._workspace(m, k)
        x = planes[s0:s0 + m]
        plan = self._rect_plan(x, m, k, st) if self._use_rect(m) else None
        if plan is not None and self.n_rect >= 2:         # both readers of the stem's output (layer 0, layer 1's shortcut) are sub-rectangle layers
            check(L.snk_stem_conv_bn_relu_f32_rect(x.data_ptr(), self.stem_w.data_ptr(), self.stem_sc.data_ptr(),
                                                   self.stem_sh.data_ptr(), a.data_ptr(), plan[3].data_ptr(), 1, m, h, w, st))
        else:
            check(L.snk_stem_conv_bn_relu_f32(x.data_ptr(), self.stem_w.data_ptr(), self.stem_sc.data_ptr(),
                                              self.stem_sh.data_ptr(), a.data_ptr(), m, h, w, st))
        cur, t1, t2 = a, b, c
        # the last layer's epilogue also does the head's 1x1 stage (snk_head_dense_f32 keeps 16 states' h1 in 64 KB of LDS)
        fused_head = self.conv_algo == "f16s" and self.blocks > 0 and self._head_fits()
        for blk in range(self.blocks):
            i0, i1 = 2 * blk, 2 * blk + 1
            self._conv(i0, cur, None, t1, m, st, plan=plan)
            if fused_head and blk == self.blocks - 1:
                h1 = self._h1_workspace(m, k)
                self._conv(i1, t1, cur, None, m, st, h1=h1)
                mk = None if mask is None else mask[s0:s0 + m]
                check(L.snk_head_dense_f32(h1.data_ptr(), self.fc1_w.data_ptr(), self.fc1_b.data_ptr(), self.fc2_w.data_ptr(),
                                           self.fc2_b.data_ptr(), 0 if mk is None else mk.data_ptr(), out[s0:s0 + m].data_ptr(),
                                           m, h, w, st))
                return
            self._conv(i1, t1, cur, t2, m, st, plan=plan)
            cur, t2 = t2, cur
        mk = None if mask is None else mask[s0:s0 + m]
        check(L.snk_head_f32(cur.data_ptr(), self.head_w.data_ptr(), self.head_s, self.head_b,
                             self.fc1_w.data_ptr(), self.fc1_b.data_ptr(), self.fc2_w.data_ptr(), self.fc2_b.data_ptr(),
                             0 if mk is None else mk.data_ptr(), out[s0:s0 + m].data_ptr(), m, h, w, st))

    def _head_fits(self):
        """snk_head_dense_f32 keeps the h1 rows of 16, 8 or 4 states in 64 KB of LDS"""
        return 4 * (self.h * self.w + 128) * 4 <= 64 * 1024

    def _h1_workspace(self, n, k=0):
        key = ("h1", k)
        if key not in self._ws or self._ws[key].shape[0] < n:
            self._ws[key] = torch.empty((n, self.h * self.w), dtype=torch.float32, device=self.device)
        return self._ws[key]

    # ---- sub-rectangle form ----------------------------------------------------------------------
    def backgrounds(self):
        """[1 + n_rect][h][w][128]: the outputs of the stem (index 0) and of the first tower layers (index 1 + i) on an
        all-background observation, computed with the full kernels themselves (so that a pixel the sub-rectangle form takes
        from here equals the pixel the full form computes, bit for bit); made again when the weights or an activation
        scale change."""
        if self._bg is None:
            m, st = 1, torch.cuda.current_stream().cuda_stream
            a16 = self.act16 is not None
            dt = self.act16 if a16 else torch.float32
            blank = torch.tensor(self.background, dtype=torch.float32, device=self.device).repeat(1, self.h, self.w, 1).contiguous()
            bufs = [torch.empty((1, self.h, self.w, 128), dtype=dt, device=self.device) for _ in range(3)]
            stem = self._fn16()[0] if a16 else self.L.snk_stem_conv_bn_relu_f32
            check(stem(blank.data_ptr(), self.stem_w.data_ptr(), self.stem_sc.data_ptr(), self.stem_sh.data_ptr(),
                       bufs[0].data_ptr(), m, self.h, self.w, st))
            bg = torch.empty((1 + self.n_rect, self.h, self.w, 128), dtype=dt, device=self.device)
            bg[0].copy_(bufs[0][0])
            cur, t1, t2 = bufs
            tm, self.conv_timing = self.conv_timing, None

            def conv(i, x, res, out):
                if a16:
                    check(self._fn16()[2](x.data_ptr(), self.conv_wT[i].data_ptr(), self.conv_sc[i].data_ptr(),
                                          self.conv_sh[i].data_ptr(), None if res is None else res.data_ptr(),
                                          out.data_ptr(), 1, m, self.h, self.w, 1, st))
                else:
                    self._conv(i, x, res, out, m, st)
            for i in range(self.n_rect):
                if i % 2 == 0:
                    conv(i, cur, None, t1)
                    bg[1 + i].copy_(t1[0])
                else:
                    conv(i, t1, cur, t2)
                    bg[1 + i].copy_(t2[0])
                    cur, t2 = t2, cur
            self.conv_timing = tm
            self._bg = bg
        return self._bg

    def _rect_args(self, i, plan, res):
        """the sub-rectangle arguments of tower layer i: descriptors, count, the producers' background images and the pixels
        they computed around the bounding box (the stem 1, tower layer j: j + 2), this layer's own image when it fills"""
        desc, counts, bg = plan[:3]
        return (desc[i].data_ptr(), counts[i].data_ptr(), bg[i].data_ptr(), i + 1,
                None if res is None else bg[i - 1].data_ptr(), i, bg[1 + i].data_ptr() if self.rect_fill[i] else None)

    def _use_rect(self, m):
        return self.n_rect > 0 and m >= self.rect_min

    def _rect_plan(self, x, m, k, st):
        """descriptors of this chunk's sub-rectangle layers: (descriptor tensor [n_rect][max_blocks][4], counts [n_rect][2])"""
        bg = self.backgrounds()
        a16 = self.act16 is not None          # the 16-bit towers cut rectangles for their own block frame
        mb = int((self.L.snk_conv_rect_max_blocks_act16 if a16 else self.L.snk_conv_rect_max_blocks)(m, self.h, self.w))
        if mb < 0:
            raise EngineError(f"sub-rectangle convolution: shape {m} x {self.h} x {self.w} not supported")
        key = ("rect", k)
        ws = None if self._ws is None else self._ws.get(key)
        if ws is None or ws[0].shape[0] != self.n_rect or ws[0].shape[1] < mb or ws[2].shape[0] < m:
            ws = (torch.empty((self.n_rect, mb, 4), dtype=torch.int32, device=self.device),
                  torch.zeros((self.n_rect, 2), dtype=torch.int32, device=self.device),
                  torch.empty((m,), dtype=torch.int32, device=self.device))
            if self._ws is None:
                self._ws = {}
            self._ws[key] = ws
        desc, counts, bbox = ws
        if self.rect_tiles is not None:          # bench.py: what the launches really executed
            counts = torch.zeros((self.n_rect, 2), dtype=torch.int32, device=self.device)
            self.rect_tiles.append((m, counts))
        grow = (C.c_int * self.n_rect)(*[i + 2 for i in range(self.n_rect)])
        # the descriptor array of layer l starts at l * max_blocks(m): the tensor may be wider (an earlier, larger chunk)
        desc_m = desc if desc.shape[1] == mb else desc.view(-1)[:self.n_rect * mb * 4].view(self.n_rect, mb, 4)
        b0, b1, b2 = self.background
        check((self.L.snk_conv_rect_plan_act16 if a16 else self.L.snk_conv_rect_plan)(
            x.data_ptr(), b0, b1, b2, m, self.h, self.w, self.n_rect, grow, bbox.data_ptr(), desc_m.data_ptr(), counts.data_ptr(), st))
        return desc_m, counts, bg, bbox

    def _conv(self, i, x, res, out, m, st, h1=None, plan=None):
        tm = self.conv_timing
        if tm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        if plan is not None and i < self.n_rect and h1 is None:
            check(self.L.snk_conv3x3_bn_f16s_rect(x.data_ptr(), self.conv_wT[i].data_ptr(), self.conv_sc[i].data_ptr(),
                                                  self.conv_sh[i].data_ptr(), None if res is None else res.data_ptr(),
                                                  out.data_ptr(), *self._rect_args(i, plan, res), m, self.h, self.w, st))
            if tm is not None:
                e1.record(torch.cuda.current_stream())
                tm.append((e0, e1, 2.0 * m * self.h * self.w * 9 * 128 * 128))
            return
        if h1 is not None:
            check(self.L.snk_conv3x3_bn_f16s_head(x.data_ptr(), self.conv_wT[i].data_ptr(), self.conv_sc[i].data_ptr(),
                                                  self.conv_sh[i].data_ptr(), None if res is None else res.data_ptr(),
                                                  None if out is None else out.data_ptr(), self.head_w.data_ptr(),
                                                  self.head_s, self.head_b, h1.data_ptr(), m, self.h, self.w, st))
            if tm is not None:
                e1.record(torch.cuda.current_stream())
                tm.append((e0, e1, 2.0 * m * self.h * self.w * 9 * 128 * 128))
            return
        fn = {"winograd": self.L.snk_conv3x3_bn_f32_winograd,
              "f16s": self.L.snk_conv3x3_bn_f16s, "f16": self.L.snk_conv3x3_bn_f16}.get(
            self.conv_algo, self.L.snk_conv3x3_bn_f32)
        check(fn(x.data_ptr(), self.conv_wT[i].data_ptr(), self.conv_sc[i].data_ptr(),
                 self.conv_sh[i].data_ptr(), None if res is None else res.data_ptr(),
                 out.data_ptr(), m, self.h, self.w, 1, st))
        if tm is not None:
            e1.record(torch.cuda.current_stream())
            tm.append((e0, e1, 2.0 * m * self.h * self.w * 9 * 128 * 128))

    def activation_report(self, planes):
        """Largest |input| of every tower layer on the given observations against the range the split-f16 kernel was
        given for it: a list of (max |x|, x_scale, headroom) with headroom = 65504 / (max |x| * x_scale).  The kernel
        clamps inputs beyond the f16 range, so headroom < 1 on representative data means the batch-norm-derived scales
        (activation_scales) do not fit these weights; use SNK_CONV_ALGO=winograd or direct for such a net."""
        assert planes.is_cuda and planes.dtype == torch.float32 and tuple(planes.shape[1:]) == self.input_shape
        m = planes.shape[0]
        st = torch.cuda.current_stream().cuda_stream
        a, b, c = self._workspace(m, 0)
        check(self.L.snk_stem_conv_bn_relu_f32(planes.contiguous().data_ptr(), self.stem_w.data_ptr(), self.stem_sc.data_ptr(),
                                               self.stem_sh.data_ptr(), a.data_ptr(), m, self.h, self.w, st))
        rep, cur, t1, t2 = [], a, b, c

        def note(x, i):
            amax = float(x[:m].abs().max().item())
            rep.append((amax, self.conv_x_scale[i], 65504.0 / max(amax * self.conv_x_scale[i], 1e-30)))
        for blk in range(self.blocks):
            note(cur, 2 * blk)
            self._conv(2 * blk, cur, None, t1, m, st)
            note(t1, 2 * blk + 1)
            self._conv(2 * blk + 1, t1, cur, t2, m, st)
            cur, t2 = t2, cur
        return rep

    def flops_per_state(self):
        hw = self.h * self.w
        return 2 * (hw * (27 * 128 + 2 * self.blocks * 9 * 128 * 128 + 128) + hw * 128 + 128 * 3)
