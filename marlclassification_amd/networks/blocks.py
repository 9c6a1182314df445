"""Parameter holders for the small networks around the LSTM cells.  Module indices inside
each ``nn.Sequential`` reproduce the reference's state-dict keys (``.0`` Linear, ``.1``
LayerNorm, ``.3`` Linear, ``.4`` LayerNorm); the arithmetic runs in libmarl_hip.so."""

from typing import Tuple

import torch as th
from torch import nn


def linear_ln_silu(n_in: int, n_out: int) -> nn.Sequential:
    """StateToFeatures layout (reference networks/state.py:14-16)."""
    return nn.Sequential(nn.Linear(n_in, n_out), nn.LayerNorm(n_out), nn.SiLU())


def mlp_two_norms(n_in: int, hidden: int, n_out: int) -> nn.Sequential:
    """MessageSender / MessageReceiver layout (reference networks/message.py:20-49)."""
    return nn.Sequential(
        nn.Linear(n_in, hidden), nn.LayerNorm(hidden), nn.SiLU(),
        nn.Linear(hidden, n_out), nn.LayerNorm(n_out), nn.SiLU(),
    )


def head(n_in: int, hidden: int, n_out: int, last: nn.Module) -> nn.Sequential:
    """Policy (last = Softmax), Critic (last = Flatten) and Prediction layouts
    (reference networks/policy.py:12-16,23-27, networks/prediction.py:11-14)."""
    return nn.Sequential(
        nn.Linear(n_in, hidden), nn.LayerNorm(hidden), nn.SiLU(), nn.Linear(hidden, n_out), last
    )


class LSTMCellWrapper(nn.Module):
    """Holder of one nn.LSTMCell's weights under the key ``_LSTMCellWrapper__lstm``
    (reference networks/recurrent.py:7-35); both cells run as one fused MFMA GEMM."""

    def __init__(self, input_size: int, n: int) -> None:
        super().__init__()
        self.__lstm = nn.LSTMCell(input_size, n)

    def forward(self, h: th.Tensor, c: th.Tensor, u: th.Tensor) -> Tuple[th.Tensor, th.Tensor]:
        raise RuntimeError("LSTM cells only run fused inside the HIP step / episode kernels")
