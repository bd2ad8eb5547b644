"""Test infrastructure: the deterministic stub Q-net of oracle/obs_key.py (stub_q) evaluated with torch ops on the
device, so that large MCTS parity tests do not pull every observation to the host.  Same integer arithmetic
(splitmix64 on the observation bit patterns, mod 2**64) and the same three float32 roundings; checked equal to
oracle.obs_key.stub_q by tests/test_mcts_gpu.py::test_device_stub_net_equals_the_oracle_stub."""
import torch


def _c(v):          # uint64 constant as the int64 with the same bit pattern
    return v - (1 << 64) if v >= (1 << 63) else v


_G, _C1, _C2 = _c(0x9E3779B97F4A7C15), _c(0xBF58476D1CE4E5B9), _c(0x94D049BB133111EB)


def _lsr(x, k):     # logical shift right of an int64 tensor
    return (x >> k) & ((1 << (64 - k)) - 1)


def _sm64(x):
    x = x + _G
    x = (x ^ _lsr(x, 30)) * _C1
    x = (x ^ _lsr(x, 27)) * _C2
    return x ^ _lsr(x, 31)


def obs_key_lo(planes):
    """planes: cuda float32 [n, h, w, 3] -> int64 [n]: the low word of the 128-bit observation key"""
    n = planes.shape[0]
    bits = planes.contiguous().view(torch.int32).reshape(n, -1, 3).to(torch.int64) & 0xFFFFFFFF
    p = torch.arange(bits.shape[1], dtype=torch.int64, device=planes.device)[None, :]
    b0, b1, b2 = bits[..., 0], bits[..., 1], bits[..., 2]
    live = ~((b0 == 0) & (b1 == 0x3F800000) & (b2 == 0))
    x = _sm64((p << 32) | b0)
    x = _sm64(x ^ ((b1 << 32) | b2))
    return torch.where(live, x, torch.zeros_like(x)).sum(dim=1)


def stub_q_device(planes, mask, chunk=131072):
    """the evaluate(planes, mask) callable DeviceMCTS expects: stub_q with the engine's obstacle mask applied
    (in chunks: the int64 temporaries of a million-row batch at BASELINE configs[2] would take tens of GB)"""
    if planes.shape[0] > chunk:
        return torch.cat([stub_q_device(planes[i:i + chunk], mask[i:i + chunk], chunk) for i in range(0, planes.shape[0], chunk)])
    klo = obs_key_lo(planes)
    cols = []
    for m in range(3):
        v = (_lsr(klo, 20 * m) if m else klo) & 0xFFFFF
        t = v.to(torch.float32) * (2.0 ** -20)
        t = t * 1.8
        cols.append(t - 0.9)
    q = torch.stack(cols, dim=1)
    return torch.where(mask.bool(), torch.full_like(q, -1.0), q).contiguous()


class DeviceStubNNet:
    def v_device(self, planes, mask):
        return stub_q_device(planes, mask)
