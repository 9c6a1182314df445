"""EpisodeSampler (reference core/episode.py): ``run_episode`` is ONE call into the HIP
library for all ``nb_step`` steps (``marl_episode_forward``); with autograd enabled the
outputs carry a single autograd node whose backward is ``marl_episode_backward`` (BPTT
through every step), so the reference's own loss code and ``loss.backward()`` work
unchanged on top of it."""

from dataclasses import dataclass
from typing import Optional, Tuple

import torch as th

from ..engine import EpisodeTensors, HipEngine
from ..fused import EpisodeDraws, draw_episode_device
from .agent import MultiAgent
from .environment import Environment


@dataclass
class EpisodeOutput:
    prediction: th.Tensor
    actions_log_probs: th.Tensor


@dataclass
class EpisodeDetailedOutput:
    step_preds: th.Tensor
    step_log_probas: th.Tensor
    step_values: th.Tensor
    step_pos: th.Tensor


class _EpisodeFunction(th.autograd.Function):
    """Autograd boundary around the fused episode: inputs are the model parameters, outputs
    step_preds / step_log_probas / step_values (+ non-differentiable positions)."""

    @staticmethod
    def forward(ctx, eng: HipEngine, img: th.Tensor, draws: EpisodeDraws, names, *params):
        # every episode owns its saved activations (a training workspace from the engine's pool), so
        # several rollouts of one model can be alive at once and (loss1 + loss2).backward() works as
        # with the reference's autograd graph (reference core/episode.py:84)
        ws = eng.train_ws_acquire()
        out = eng.episode_forward(img, draws.pos0, draws.h0, draws.c0, draws.hc0, draws.cc0,
                                  draws.noise, None, True, rng=draws.rng, ws=ws)
        ctx.eng, ctx.ws, ctx.img = eng, ws, img
        ctx.cfg_key = eng._cfg_key
        ctx.pack_generation = eng.pack_generation
        ctx.names = names
        ctx.shapes = [p.shape for p in params]
        ctx.mark_non_differentiable(out.step_pos, out.step_actions)
        return out.step_preds, out.step_log_probas, out.step_values, out.step_pos, out.step_actions

    @staticmethod
    def backward(ctx, g_preds, g_logp, g_values, _g_pos, _g_act):
        eng: HipEngine = ctx.eng
        if ctx.ws is None:
            raise RuntimeError("this episode's saved activations were already released by an earlier "
                               "backward (run the episode again instead of retain_graph)")
        if ctx.pack_generation != eng.pack_generation:
            raise RuntimeError("the model's weights were modified (re-packed) after this episode's rollout: "
                               "its backward needs the weights the rollout used - call backward before "
                               "the optimiser step")
        if ctx.cfg_key != eng._cfg_key:  # another shape ran in between: switch the engine back
            na, nb, ns, shape, u8 = ctx.cfg_key
            eng.configure(na, nb, ns, shape, img_u8=u8)
        grads = {k: th.empty(s, device=eng.device) for k, s in zip(ctx.names, ctx.shapes)}
        eng.episode_backward(g_preds, g_logp, g_values, grads, ws=ctx.ws, img=ctx.img)
        eng.train_ws_release(ctx.ws)
        ctx.ws = None
        return (None, None, None, None) + tuple(grads[k] for k in ctx.names)


class EpisodeSampler:
    def __init__(self, agents: MultiAgent, env: Environment, nb_step: int, stream_id: int = 1) -> None:
        self.__agents = agents
        self.__env = env
        self.__nb_step = nb_step
        # parity hook: when set, these draws replace the random ones (tests inject the
        # reference's host-drawn positions / states / noise)
        self.fixed_draws: Optional[EpisodeDraws] = None
        # perf mode (default): every draw comes from the library's counter-based generator, keyed
        # by torch's seed (th.manual_seed keeps runs reproducible) and an episode counter.  False:
        # torch draws in the reference's order (positions, h, c, h^, c^, per-step Exp(1)).
        self.device_rng = True
        self.__episodes = 0
        self.__rng_seed: Optional[int] = None
        # every sampler is its own stream of the generator: the key mixes torch's seed with the
        # rank (shards draw different positions / states / noise under the usual identical
        # th.manual_seed on all ranks) and `stream_id` - an explicit argument, not a
        # construction counter: same seed, same draws, whatever
        # else the process built before
        self.__stream_id = int(stream_id)

    @property
    def nb_step(self) -> int:
        return self.__nb_step

    @property
    def agents(self) -> MultiAgent:
        return self.__agents

    @property
    def env(self) -> Environment:
        return self.__env

    def draw_key(self, seed: int) -> int:
        """Generator key of this sampler's draws: (torch seed, rank, sampler id)."""
        import os

        from ..parallel import shard_seed

        rank = int(os.environ.get("RANK", "0"))
        return (shard_seed(seed & 0xFFFFFFFFFFFF, rank) * 1_000_003 + self.__stream_id) & ((1 << 63) - 1)

    def prepare(self, img_batch: th.Tensor) -> Tuple[HipEngine, th.Tensor, EpisodeDraws]:
        """Everything before the kernels: device transfer, engine configuration, weight
        packing, and the reference's random draws in the reference's order (positions,
        h, c, h^, c^, per-step Exp(1) noise)."""
        agents, env = self.__agents, self.__env
        model = agents.model
        device = agents.device
        img = img_batch.to(device)
        na, nb, ns = len(agents), img.shape[0], self.__nb_step
        eng = model.hip_engine(env.actions)
        # uint8 batches ([Nb,C,H,W], 0..255) stay uint8: ToTensor happens inside the gather kernel
        eng.configure(na, nb, ns, img.shape[1:], img_u8=img.dtype == th.uint8)
        model.ensure_packed(eng)
        if self.fixed_draws is not None:
            env.place(img, na, positions=self.fixed_draws.pos0)
            return eng, img, self.fixed_draws
        if self.device_rng:
            seed = th.initial_seed()
            if seed != self.__rng_seed:  # th.manual_seed() restarts the episode counter
                self.__rng_seed, self.__episodes = seed, 0
            d = draw_episode_device(eng, self.draw_key(seed), self.__episodes)
            self.__episodes += 1
            env.place(img, na, positions=d.pos0)
            return eng, img, d
        pos0 = env.place(img, na)
        st = model.random_first_state(na, nb)
        noise = th.empty(ns, na, nb, env.nb_actions, device=device).exponential_(1.0)
        return eng, img, EpisodeDraws(pos0, st.h, st.c, st.h_caret, st.c_caret, noise)

    def __episode_impl(self, img_batch: th.Tensor) -> EpisodeDetailedOutput:
        eng, img, draws = self.prepare(img_batch)
        model = self.__agents.model
        if th.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
            named = list(model.named_parameters())
            names = tuple(k for k, _ in named)
            preds, logp, values, pos, _ = _EpisodeFunction.apply(
                eng, img, draws, names, *[p for _, p in named])
        else:
            out = eng.episode_forward(img, draws.pos0, draws.h0, draws.c0, draws.hc0, draws.cc0,
                                      draws.noise, None, False, rng=draws.rng)
            preds, logp, values, pos = (out.step_preds, out.step_log_probas, out.step_values,
                                        out.step_pos)
        self.__env._set_positions(pos[-1])
        return EpisodeDetailedOutput(preds, logp, values, pos)

    def run_episode(self, img_batch: th.Tensor) -> EpisodeDetailedOutput:
        return self.__episode_impl(img_batch)

    def run_episode_get_last_step(self, img_batch: th.Tensor) -> EpisodeOutput:
        out = self.__episode_impl(img_batch)
        return EpisodeOutput(prediction=out.step_preds[-1],
                             actions_log_probs=out.step_log_probas[-1])

    def run_episode_raw(self, img_batch: th.Tensor, train: bool,
                        draws: Optional[EpisodeDraws] = None) -> Tuple[HipEngine, EpisodeTensors]:
        """No autograd node: used by the fused Trainer (loss + backward are HIP calls)."""
        eng, img, d = self.prepare(img_batch)
        if draws is not None:
            d = draws
        out = eng.episode_forward(img, d.pos0, d.h0, d.c0, d.hc0, d.cc0, d.noise, None, train,
                                  rng=d.rng)
        self.__env._set_positions(out.step_pos[-1])
        return eng, out
