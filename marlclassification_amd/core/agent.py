"""MultiAgent: recurrent state + policy sampling around ModelsWrapper (reference
core/agent.py).  ``act`` is one ``marl_step_forward`` call that also samples
(argmax(p / q), q ~ Exp(1): what th.multinomial(p, 1) computes) and returns log p[a]."""

from dataclasses import dataclass

import torch as th

from ..networks.models import ModelsWrapper, RecurrentOutput


@dataclass
class AgentOutput:
    actions: th.Tensor
    actions_log_probs: th.Tensor
    predictions: th.Tensor
    values: th.Tensor


class MultiAgent:
    def __init__(self, nb_agents: int, model: ModelsWrapper, stream_id: int = 1) -> None:
        self.__nb_agents = nb_agents
        self.__model = model
        self.__hidden: RecurrentOutput | None = None
        self.__last_msg: th.Tensor | None = None
        # parity hook: a [Na, Nb, nA] tensor of Exp(1) draws used by the NEXT act() instead of
        # the generator (th.multinomial(p, 1) == argmax(p / q))
        self.fixed_noise: th.Tensor | None = None
        # default: the sampling kernel draws its own variates (library generator keyed by torch's
        # seed and a call counter); False: torch draws the Exp(1) tensor
        self.device_rng = True
        self.__calls = 0
        # generator stream of this object's draws (mixed with torch's seed and the rank, see act).  An
        # explicit argument, not a per-process construction counter: the draws of a run depend on the
        # seed alone, not on how many agents a test / notebook / resume built before this one.
        self.__stream_id = int(stream_id)

    def reset(self, batch_size: int) -> None:
        self.__hidden = self.__model.random_first_state(len(self), batch_size)
        self.__last_msg = self.__model.zero_first_message(len(self), batch_size)

    def act(self, observation: th.Tensor, norm_pos: th.Tensor) -> AgentOutput:
        if self.__hidden is None:
            self.reset(observation.shape[1])
        model = self.__model
        na, nb = observation.shape[:2]
        eng = model.hip_engine(None)
        eng.configure(na, nb, 1, (observation.shape[2], observation.shape[3] + 1,
                                  observation.shape[4] + 1))
        model.ensure_packed(eng)
        hid = self.__hidden
        noise, rng = None, None
        if self.fixed_noise is not None:
            noise, self.fixed_noise = self.fixed_noise, None
        elif self.device_rng:
            import os

            from ..parallel import shard_seed

            key = shard_seed(th.initial_seed() & 0xFFFFFFFFFFFF, int(os.environ.get("RANK", "0")))
            key = (key * 1_000_003 + (1 << 20) + self.__stream_id) & ((1 << 63) - 1)
            rng = (key, (1 << 40) + self.__calls)  # offsets apart from the episodes'
            self.__calls += 1
        else:
            noise = th.empty(na, nb, model.nb_action, device=observation.device).exponential_(1.0)
        probs, values, preds, msg, h, c, hc, cc, actions, logp = eng.step_forward(
            observation, self.__last_msg, norm_pos, hid.h, hid.c, hid.h_caret, hid.c_caret, noise,
            rng=rng)
        self.__hidden = RecurrentOutput(h, c, hc, cc)
        self.__last_msg = msg
        return AgentOutput(actions=actions, actions_log_probs=logp, predictions=preds, values=values)

    @property
    def model(self) -> ModelsWrapper:
        return self.__model

    @property
    def nb_class(self) -> int:
        return self.__model.nb_class

    @property
    def device(self) -> th.device:
        return self.__model.device

    def __len__(self) -> int:
        return self.__nb_agents
