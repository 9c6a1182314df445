"""Feature extractors b_theta5 (reference: networks/vision.py).  These classes only HOLD the
parameters - with the reference's state-dict keys - and describe the conv stack; the
convolutions, GroupNorm and SiLU run inside libmarl_hip.so (csrc/cnn.hip, gemm.hip,
rowops.hip).  There is no torch forward."""

from abc import ABC, abstractmethod
from typing import List, Sequence, Tuple

import torch as th
from torch import nn


class VisionCnnModule(nn.Module, ABC):
    """Interface of networks/vision.py:9-17 plus what the HIP engine needs to know."""

    @property
    @abstractmethod
    def out_size(self) -> int: ...

    @property
    @abstractmethod
    def hip_name(self) -> str:
        """Key into engine.CNN_SPECS."""

    @property
    @abstractmethod
    def window(self) -> int: ...

    def forward(self, o_t: th.Tensor) -> th.Tensor:
        raise RuntimeError(
            "feature extractors have no standalone forward in this package: the CNN is fused "
            "into the HIP episode / step kernels (use ModelsWrapper or EpisodeSampler)"
        )


class _Generic2dCnnModule(VisionCnnModule):
    """L x [Conv2d k3 s2 p1 -> GroupNorm -> SiLU] -> Flatten, as parameter holder
    (reference networks/vision.py:23-52; keys ``__layers.{3l}`` / ``{3l+1}``)."""

    def __init__(self, f: int, channels: Sequence[int], groups: Sequence[int], name: str) -> None:
        super().__init__()
        blocks: List[nn.Module] = []
        size = f
        for c_in, c_out, g in zip(channels[:-1], channels[1:], groups):
            blocks += [nn.Conv2d(c_in, c_out, 3, 2, 1), nn.GroupNorm(g, c_out), nn.SiLU()]
            size = (size - 1) // 2 + 1
        blocks.append(nn.Flatten(1, -1))
        self.__layers = nn.Sequential(*blocks)
        self.__f = f
        self.__name = name
        self.__out_size = channels[-1] * size * size

    @property
    def out_size(self) -> int:
        return self.__out_size

    @property
    def hip_name(self) -> str:
        return self.__name

    @property
    def window(self) -> int:
        return self.__f


def _spec(name: str) -> Tuple[List[int], List[int]]:
    from ..engine import CNN_SPECS

    return CNN_SPECS[name]


class MnistCnn(_Generic2dCnnModule):
    """1->8->16, reads channel 0 only (reference vision.py:55-65)."""

    def __init__(self, f: int) -> None:
        super().__init__(f, *_spec("mnist"), "mnist")


class Resisc45Cnn(_Generic2dCnnModule):
    def __init__(self, f: int) -> None:
        super().__init__(f, *_spec("resisc45"), "resisc45")


class AidCnn(_Generic2dCnnModule):
    def __init__(self, f: int) -> None:
        super().__init__(f, *_spec("aid"), "aid")


class WorldStratCnn(_Generic2dCnnModule):
    def __init__(self, f: int) -> None:
        super().__init__(f, *_spec("worldstrat"), "worldstrat")


class SkinCancerCnn(_Generic2dCnnModule):
    def __init__(self, f: int) -> None:
        super().__init__(f, *_spec("skin_cancer"), "skin_cancer")


CNN_BY_NAME = {
    "mnist": MnistCnn,
    "resisc45": Resisc45Cnn,
    "aid": AidCnn,
    "worldstrat": WorldStratCnn,
    "skin_cancer": SkinCancerCnn,
}
