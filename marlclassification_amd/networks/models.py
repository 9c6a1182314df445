"""ModelsWrapper: the reference's bundle of agent networks (networks/models.py:31-162) as a
drop-in ``nn.Module`` whose arithmetic is libmarl_hip.so.

* same constructor arguments, same ``state_dict()`` keys and shapes (name-mangled private
  attributes), so reference checkpoints load and checkpoints written here load there;
* parameters are views into one flat fp32 buffer (``flat_state``) so Adam and the
  data-parallel all-reduce are single kernels / collectives;
* ``forward`` is the reference's per-step network (MultiAgent.act uses it); whole
  episodes go through core.episode.EpisodeSampler (one call for all steps).
"""

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch as th
from torch import nn

from ..engine import HipEngine, ModelSpec
from ..fused import FlatParams
from .blocks import LSTMCellWrapper, head, linear_ln_silu, mlp_two_norms
from .init import init_layers
from .vision import VisionCnnModule


@dataclass
class ModelOutput:
    actions_probabilities: th.Tensor
    values: th.Tensor
    predictions: th.Tensor
    messages: th.Tensor


@dataclass
class RecurrentOutput:
    h: th.Tensor
    c: th.Tensor
    h_caret: th.Tensor
    c_caret: th.Tensor


class ModelsWrapper(nn.Module):
    def __init__(
        self,
        ft_extractor: VisionCnnModule,
        n_b: int,
        n_a: int,
        n_m: int,
        n_m_o: int,
        n_d: int,
        d: int,
        nb_action: int,
        nb_class: int,
        hidden_size_belief: int,
        hidden_size_action: int,
    ) -> None:
        super().__init__()
        if d != 2:
            raise ValueError("the HIP path implements 2-D images only (state_dim == 2)")
        n_in = ft_extractor.out_size + n_d + n_m_o

        self.__map_obs = ft_extractor
        self.__map_pos = linear_ln_silu(d, n_d)
        self.__encode_msg = mlp_two_norms(n_b, 2 * n_m, n_m)
        self.__decode_msg = mlp_two_norms(n_m, 2 * n_m, n_m_o)
        self.__belief_unit = LSTMCellWrapper(n_in, n_b)
        self.__action_unit = LSTMCellWrapper(n_in, n_a)
        self.__policy = head(n_a, hidden_size_action, nb_action, nn.Softmax(dim=-1))
        self.__critic = head(n_a, hidden_size_action, 1, nn.Flatten(-2, -1))
        self.__predict = head(n_b, hidden_size_belief, nb_class, nn.Identity())

        self.__dims = dict(n_b=n_b, n_a=n_a, n_m=n_m, n_m_o=n_m_o, n_d=n_d, nb_class=nb_class,
                           nlb=hidden_size_belief, nla=hidden_size_action)
        self.__nb_action = nb_action
        self.apply(init_layers)

        self.__flat: Optional[FlatParams] = None
        self.__engines: Dict[Tuple, HipEngine] = {}
        self.__packed_token: Dict[int, Tuple] = {}

    # ---- reference surface -------------------------------------------------------------
    @property
    def nb_class(self) -> int:
        return self.__dims["nb_class"]

    @property
    def nb_action(self) -> int:
        return self.__nb_action

    @property
    def device(self) -> th.device:
        return next(self.parameters()).device

    def random_first_state(self, nb_agents: int, batch_size: int) -> RecurrentOutput:
        """h, c, h^, c^ ~ N(0, 1), drawn in that order (reference models.py:148-159)."""
        dev = self.device
        n_b, n_a = self.__dims["n_b"], self.__dims["n_a"]
        return RecurrentOutput(
            h=th.randn(nb_agents, batch_size, n_b, device=dev),
            c=th.randn(nb_agents, batch_size, n_b, device=dev),
            h_caret=th.randn(nb_agents, batch_size, n_a, device=dev),
            c_caret=th.randn(nb_agents, batch_size, n_a, device=dev),
        )

    def zero_first_message(self, nb_agents: int, batch_size: int) -> th.Tensor:
        return th.zeros(nb_agents, batch_size, self.__dims["n_m"], device=self.device)

    def forward(
        self,
        img_patch: th.Tensor,
        msg_t: th.Tensor,
        norm_pos: th.Tensor,
        recurrent_hidden: RecurrentOutput,
    ) -> Tuple[ModelOutput, RecurrentOutput]:
        """One step of every network (reference models.py:78-138) through
        ``marl_step_forward``.  Inference only: gradients flow through whole episodes
        (EpisodeSampler), not through single steps."""
        na, nb = img_patch.shape[:2]
        eng = self.hip_engine(None)
        eng.configure(na, nb, 1, (img_patch.shape[2], img_patch.shape[3] + 1, img_patch.shape[4] + 1))
        self.ensure_packed(eng)
        rh = recurrent_hidden
        probs, values, preds, msg, h, c, hc, cc = eng.step_forward(
            img_patch, msg_t, norm_pos, rh.h, rh.c, rh.h_caret, rh.c_caret)
        return ModelOutput(probs, values, preds, msg), RecurrentOutput(h, c, hc, cc)

    # ---- HIP plumbing --------------------------------------------------------------------
    def model_spec(self, actions: Optional[List[List[int]]]) -> ModelSpec:
        cnn = self.__map_obs
        if actions is None:
            actions = [[0, 0]] * self.__nb_action
        if len(actions) != self.__nb_action:
            raise ValueError(f"model has {self.__nb_action} actions, environment {len(actions)}")
        return ModelSpec(ft_extr=cnn.hip_name, window=cnn.window, actions=[list(a) for a in actions],
                         **self.__dims)

    def hip_engine(self, actions: Optional[List[List[int]]]) -> HipEngine:
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(
                "ModelsWrapper runs on the GPU only: move it with .to('cuda') (the HIP library "
                "is the only implementation; there is no CPU fallback)"
            )
        key = (str(dev), None if actions is None else tuple(map(tuple, actions)))
        eng = self.__engines.get(key)
        if eng is None:
            eng = HipEngine(self.model_spec(actions), dev)
            self.__engines[key] = eng
        return eng

    def flat_state(self) -> FlatParams:
        """Flat parameter / gradient / Adam buffers; the nn.Parameters become views into the
        flat parameter buffer (re-done transparently after .to(device))."""
        dev = self.device
        named = list(self.named_parameters())
        flat = self.__flat
        ok = flat is not None and flat.params.device == dev
        if ok:
            views = flat.param_views()
            ok = all(p.data_ptr() == views[k].data_ptr() for k, p in named)
        if not ok:
            fresh = FlatParams({k: tuple(p.shape) for k, p in named}, dev)
            views = fresh.param_views()
            with th.no_grad():
                for k, p in named:
                    views[k].copy_(p.data)
                    p.data = views[k]
            if flat is not None and flat.params.device == dev and flat.numel == fresh.numel:
                fresh.exp_avg.copy_(flat.exp_avg)
                fresh.exp_avg_sq.copy_(flat.exp_avg_sq)
                fresh.step = flat.step
            self.__flat = fresh
            self.__packed_token = {}
        return self.__flat

    def _version_token(self) -> Tuple:
        return tuple(p._version for p in self.parameters())

    def ensure_packed(self, eng: HipEngine) -> None:
        """Refresh the engine's padded / transposed weight copies if any parameter changed
        through torch (load_state_dict, an optimiser, manual edits)."""
        flat = self.flat_state()
        token = self._version_token() + (eng.weights_token(),)
        if self.__packed_token.get(id(eng)) != token:
            eng.pack(flat.param_views())
            self.__packed_token[id(eng)] = token

    def mark_updated(self, eng: HipEngine) -> None:
        """Called after the HIP Adam kernel wrote the flat buffer (no torch version bump)."""
        eng.pack(self.flat_state().param_views())
        self.__packed_token = {id(eng): self._version_token() + (eng.weights_token(),)}
