"""Initial weights: the reference's recipe (networks/init.py:6-29) applied to the
parameter holders - orthogonal matrices with gain sqrt(2), zero biases, unit norm scales.
Host-side, one-off; parity of initial weights only."""

import math

from torch import nn

_GAIN = math.sqrt(2.0)


def init_layers(module: nn.Module) -> None:
    matrices, zeros, ones = [], [], []
    if isinstance(module, (nn.Linear, nn.Conv2d)):
        matrices.append(module.weight)
        if module.bias is not None:
            zeros.append(module.bias)
    elif isinstance(module, nn.LSTMCell):
        # weight_hh is drawn before weight_ih, as in the reference
        matrices += [module.weight_hh, module.weight_ih]
        if module.bias:
            zeros += [module.bias_hh, module.bias_ih]
    elif isinstance(module, (nn.LayerNorm, nn.GroupNorm)):
        if module.weight is not None:
            ones.append(module.weight)
            zeros.append(module.bias)
    for w in matrices:
        nn.init.orthogonal_(w, gain=_GAIN)
    for b in zeros:
        nn.init.zeros_(b)
    for g in ones:
        nn.init.ones_(g)
