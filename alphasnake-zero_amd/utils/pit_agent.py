"""Drop-in mirror of the reference's ``utils.pit_agent`` (pit_agent.py): greedy agent, nnet.v + argmax
(SURVEY.md section 8 row f-2).  ``make_moves`` takes the list of observations the reference passes or, from
the engine-backed pit runner, the device tensor of observation planes (+ obstacle mask)."""
import numpy as np
import torch

from snake_engine._lib import lib, check


class Agent:

    def __init__(self, nnet, game_and_snake_cnt=None):
        self.nnet = nnet
        self.game_and_snake_cnt = game_and_snake_cnt

    def make_moves(self, states, ids=None, mask=None):
        if isinstance(states, torch.Tensor) and hasattr(self.nnet, "v_device"):
            V = self.nnet.v_device(states, mask)
            return self._argmax_device(V)
        if isinstance(states, torch.Tensor):
            states = list(states.cpu().numpy())
        if len(states) == 0:
            return []
        V = self.nnet.v(states)
        return self.argmaxs(V)

    @staticmethod
    def _argmax_device(V):
        n = V.shape[0]
        if n == 0:
            return []
        pmf = torch.empty_like(V)
        am = torch.empty((n,), dtype=torch.uint8, device=V.device)
        check(lib().snk_softermax_argmax(V.contiguous().data_ptr(), n, 2.0, pmf.data_ptr(), am.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream))
        return am.cpu().numpy().astype(int).tolist()

    def argmaxs(self, Z):
        """pit_agent.py:15-28: strict '>' comparisons, later index wins ties"""
        Z = np.asarray(Z, np.float32).reshape(-1, 3)
        if torch.cuda.is_available():
            return self._argmax_device(torch.as_tensor(Z, device="cuda"))
        raise RuntimeError("pit_agent.Agent.argmaxs needs the GPU library (no CPU fallback)")
