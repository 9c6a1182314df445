"""Environment: image batch, agent positions, action semantics (reference
core/environment.py).  Same public surface; the crop and the bounded move are HIP kernels
(``marl_patch_gather``: O(f*f) gather instead of the reference's O(H*W) mask +
masked_select; ``marl_transition``).  Inside ``EpisodeSampler`` none of these methods is
called per step - the fused episode keeps positions on the device."""

from typing import List, Optional

import torch as th

from .. import engine as _eng


class Environment:
    def __init__(self, actions: List[List[int]], window_size: int) -> None:
        self.__actions = [list(a) for a in actions]
        self.__window_size = window_size
        self.__img_batch: th.Tensor = th.empty([1])
        self.__img_sizes: List[int] = []
        self.__pos = th.empty(0)

    def reset(self, img_batch: th.Tensor, nb_agents: int) -> th.Tensor:
        """Places agents uniformly at random (one randint per spatial dim, H first, as
        reference environment.py:33-43) and returns the first observation."""
        self.place(img_batch, nb_agents)
        return self.observe()

    def place(self, img_batch: th.Tensor, nb_agents: int,
              positions: Optional[th.Tensor] = None) -> th.Tensor:
        """reset() without the observation gather (what the fused episode needs).
        ``positions`` ([Na, Nb, 2] int64): use these instead of drawing (parity / device RNG)."""
        if img_batch.dim() != 4:
            raise ValueError("expected an image batch [Nb, C, H, W]")
        self.__img_batch = img_batch
        self.__img_sizes = list(img_batch.shape[2:])
        batch = img_batch.shape[0]
        if positions is not None:
            self.__pos = positions
            return self.__pos
        self.__pos = th.stack(
            [th.randint(s - self.__window_size, (nb_agents, batch), device=img_batch.device)
             for s in self.__img_sizes],
            dim=-1,
        )
        return self.__pos

    def observe(self) -> th.Tensor:
        assert self.__img_sizes, "reset() must be called before observe()"
        return _eng.patch_gather(self.__img_batch, self.__pos, self.__window_size)

    def step(self, action_indices: th.Tensor) -> th.Tensor:
        assert self.__img_sizes, "reset() must be called before step()"
        self.__pos = _eng.transition(self.__pos, action_indices, self.__actions,
                                     self.__img_sizes, self.__window_size)
        return self.observe()

    def _set_positions(self, pos: th.Tensor) -> None:
        self.__pos = pos

    @property
    def positions(self) -> th.Tensor:
        return self.__pos

    @property
    def normalized_positions(self) -> th.Tensor:
        return _eng.normalize_positions(self.__pos, self.__img_sizes)

    @property
    def window_size(self) -> int:
        return self.__window_size

    @property
    def actions(self) -> List[List[int]]:
        return self.__actions

    @property
    def nb_actions(self) -> int:
        return len(self.__actions)
