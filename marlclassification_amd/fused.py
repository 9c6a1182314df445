"""One A2C training iteration entirely on the HIP path (the body of the reference's
``Trainer.train_epoch`` loop, training/trainer.py:66-116): rollout -> loss + output
gradients -> backward through the episode -> [gradient all-reduce] -> Adam -> re-pack.

Parameters live in ONE flat fp32 buffer (the model's ``nn.Parameter``s are views into it),
gradients in a second one, so that Adam is a single kernel and data parallelism is a single
all-reduce (parallel.py).  No host synchronisation happens inside ``iteration``.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch as th

from .engine import EpisodeTensors, HipEngine, ModelSpec


@dataclass
class EpisodeDraws:
    """The reference's random draws for one episode, in its draw order:
    positions (environment.py:33-43), h, c, h^, c^ (models.py:148-159), then one Exp(1)
    tensor per step (``th.multinomial`` == argmax(p / q), agent.py:53-55)."""

    pos0: th.Tensor
    h0: th.Tensor
    c0: th.Tensor
    hc0: th.Tensor
    cc0: th.Tensor
    noise: Optional[th.Tensor]  # None: the sampling kernel draws its own Exp(1) variates (`rng`)
    rng: Optional[Tuple[int, int]] = None  # (seed, offset) of the library's counter-based generator


def draw_episode(spec: ModelSpec, na: int, nb: int, ns: int, sizes, device,
                 generator: Optional[th.Generator] = None) -> EpisodeDraws:
    """Device-side draws (torch's Philox generator: RNG plumbing, not compute)."""
    f = spec.window
    pos0 = th.stack(
        [th.randint(int(s) - f, (na, nb), device=device, generator=generator) for s in sizes],
        dim=-1,
    )
    h0 = th.randn(na, nb, spec.n_b, device=device, generator=generator)
    c0 = th.randn(na, nb, spec.n_b, device=device, generator=generator)
    hc0 = th.randn(na, nb, spec.n_a, device=device, generator=generator)
    cc0 = th.randn(na, nb, spec.n_a, device=device, generator=generator)
    noise = th.empty(ns, na, nb, len(spec.actions), device=device).exponential_(
        1.0, generator=generator
    )
    return EpisodeDraws(pos0, h0, c0, hc0, cc0, noise)


def draw_episode_device(engine: HipEngine, seed: int, offset: int) -> EpisodeDraws:
    """Perf mode: every draw of the episode comes from the library (one launch for positions
    and initial states; the per-step Exp(1) noise is drawn inside the sampling kernel)."""
    pos0, h0, c0, hc0, cc0, _ = engine.draw_episode(seed, offset)
    return EpisodeDraws(pos0, h0, c0, hc0, cc0, None, (seed, offset))


class FlatParams:
    """Flat fp32 parameter / gradient / Adam-moment buffers with per-tensor views."""

    def __init__(self, shapes: Dict[str, Tuple[int, ...]], device: th.device) -> None:
        self.names: List[str] = list(shapes)
        self.shapes = dict(shapes)
        self.offsets: Dict[str, int] = {}
        off = 0
        for k, s in shapes.items():
            self.offsets[k] = off
            n = 1
            for d in s:
                n *= d
            # 16-byte aligned slices so every tensor can be read with 128-bit loads
            off += (n + 3) & ~3
        self.numel = off
        self.params = th.zeros(off, device=device)
        self.grads = th.zeros(off, device=device)
        self.exp_avg = th.zeros(off, device=device)
        self.exp_avg_sq = th.zeros(off, device=device)
        self.step = 0

    def _views(self, flat: th.Tensor) -> Dict[str, th.Tensor]:
        out = {}
        for k, s in self.shapes.items():
            n = 1
            for d in s:
                n *= d
            out[k] = flat[self.offsets[k]: self.offsets[k] + n].view(s)
        return out

    def param_views(self) -> Dict[str, th.Tensor]:
        return self._views(self.params)

    def grad_views(self) -> Dict[str, th.Tensor]:
        return self._views(self.grads)

    def load(self, tensors: Dict[str, th.Tensor]) -> None:
        views = self.param_views()
        with th.no_grad():
            for k in self.names:
                views[k].copy_(tensors[k])


class FusedA2C:
    """rollout + loss + backward + Adam on one GPU; ``allreduce`` hooks in data parallelism."""

    def __init__(self, engine: HipEngine, flat: FlatParams, lr: float, gamma: float,
                 allreduce: Optional[Callable[[th.Tensor], float]] = None,
                 use_graph: bool = False) -> None:
        if use_graph and allreduce is not None:
            raise ValueError("hipGraph replay covers the single-GPU iteration (no collective inside)")
        self._graph = None  # (key, exec handle, stream, persistent tensors)
        self.engine = engine
        self.flat = flat
        self.lr = lr
        self.gamma = gamma
        self.allreduce = allreduce
        self._pviews = flat.param_views()
        self._gviews = flat.grad_views()
        self._loss_bufs = None
        self._packed_gen = -1  # engine.weights_generation at the last pack (-1: never packed)

    def pack(self) -> None:
        self.engine.pack(self._pviews)
        self._packed_gen = self.engine.weights_generation

    def rollout(self, img: th.Tensor, draws: EpisodeDraws, train: bool,
                forced_actions: Optional[th.Tensor] = None) -> EpisodeTensors:
        if self._packed_gen != self.engine.weights_token():  # never packed, or the workspace was re-laid-out
            self.pack()
        return self.engine.episode_forward(img, draws.pos0, draws.h0, draws.c0, draws.hc0,
                                           draws.cc0, draws.noise, forced_actions, train,
                                           rng=draws.rng)

    def iteration_graph(self, img: th.Tensor, y: th.Tensor, seed: int,
                        offset: int) -> Tuple[EpisodeTensors, th.Tensor]:
        """The same iteration as ONE hipGraph replay (launch-bound shapes: hundreds of small
        kernels per iteration).  The first call runs one eager iteration (one-off set-up inside
        the library), then captures draw -> rollout -> loss -> backward -> Adam -> re-pack ->
        counter tick on a side stream; later calls replay it.  What changes between iterations
        lives on the device (generator offset, Adam step: the counter block), ``img`` / ``y`` are
        read from the tensors of the capturing call (same storage every call, or re-capture).
        Outputs are persistent tensors, overwritten by every replay."""
        eng = self.engine
        from . import engine as _engine_mod

        # (the tune epoch: after engine.tune() the workspaces baked into a captured graph are freed /
        # re-laid-out - the graph must be captured again)
        key = (img.data_ptr(), y.data_ptr(), eng._cfg_key, seed, _engine_mod._tune_epoch)
        if self._graph is None or self._graph[0] != key:
            if self._graph is not None:
                eng.lib.marl_graph_destroy(self._graph[1])
                self._graph = None
            # this call runs eagerly (one-off set-up inside the library happens here) ...
            eager = self.iteration(img, y, draw_episode_device(eng, seed, offset))
            offset += 1  # ... and the graph captured below starts at the NEXT iteration
            stream = th.cuda.Stream(device=eng.device)
            out = eng.new_outputs()
            draws = eng.draw_episode(seed, 0)  # persistent draw tensors
            cnt = eng.new_counters()
            gviews = self._gviews
            stream.wait_stream(th.cuda.current_stream(eng.device))
            with th.cuda.stream(stream):
                eng.counters_set(cnt, offset, self.flat.step + 1, self.lr)
                th.cuda.synchronize(eng.device)
                eng.graph_begin()
                try:
                    eng.draw_episode(seed, 0, into=draws, counters=cnt)
                    eng.episode_forward(img, draws[0], draws[1], draws[2], draws[3], draws[4], None,
                                        None, True, rng=(seed, 0), out=out, counters=cnt)
                    gp, gl, gv, scalars, _ = eng.a2c_loss(out, y, self.gamma, 0, self._loss_bufs)
                    eng.episode_backward(gp, gl, gv, gviews)
                    eng.adam(self.flat.params, self.flat.grads, self.flat.exp_avg,
                             self.flat.exp_avg_sq, 1, self.lr, counters=cnt)
                    eng.pack(self._pviews)
                    eng.counters_tick(cnt, self.lr)
                except BaseException:
                    # the capture is invalid now: end it, drop whatever that reports, and let the
                    # ROOT cause propagate (no stale graph is left behind: self._graph is None)
                    try:
                        eng.graph_end()
                    except Exception:
                        pass
                    raise
                handle = eng.graph_end()
            # what the device-side counter block holds after the capture: the state the FIRST replay
            # expects (generator offset, optimiser step, learning rate)
            self._graph = (key, handle, stream, (out, draws, cnt, scalars),
                           {"offset": offset, "step": self.flat.step + 1, "cap_lr": self.lr})
            return eager
        _, handle, stream, (out, _draws, cnt, scalars), expect = self._graph
        stream.wait_stream(th.cuda.current_stream(eng.device))
        with th.cuda.stream(stream):
            # eager iterations in between, another offset or a changed learning rate: the device
            # counters are re-synchronised with the host's view before the replay
            # (the tick at the end of the captured graph recomputes the step size with the learning
            # rate of the CAPTURE: after a change of self.lr the counters are set before every replay)
            if (expect["offset"] != offset or expect["step"] != self.flat.step + 1 or
                    expect["cap_lr"] != self.lr):
                eng.counters_set(cnt, offset, self.flat.step + 1, self.lr)
            eng.graph_launch(handle)
        th.cuda.current_stream(eng.device).wait_stream(stream)
        self.flat.step += 1
        eng.fwd_generation += 1
        expect.update(offset=offset + 1, step=self.flat.step + 1)
        return out, scalars

    def iteration(self, img: th.Tensor, y: th.Tensor, draws: EpisodeDraws) -> Tuple[EpisodeTensors, th.Tensor]:
        """Returns the episode outputs and the device tensor {loss, path, error, critic}."""
        eng = self.engine
        out = self.rollout(img, draws, True)
        if self._loss_bufs is None or self._loss_bufs[0].shape != out.step_preds.shape:
            dev = eng.device
            self._loss_bufs = (
                th.empty_like(out.step_preds), th.empty_like(out.step_log_probas),
                th.empty_like(out.step_values), th.zeros(4, device=dev),
                th.zeros(3, dtype=th.float64, device=dev),
            )
        gp, gl, gv, scalars, _ = eng.a2c_loss(out, y, self.gamma, 0, self._loss_bufs)
        bucketed = hasattr(self.allreduce, "before_backward")
        if bucketed:  # (two buckets: the heads' slice leaves while the reverse loop still runs)
            self.allreduce.before_backward(eng)
        try:
            eng.episode_backward(gp, gl, gv, self._gviews)
        finally:
            if bucketed:
                self.allreduce.after_backward(eng)
        scale = 1.0
        if self.allreduce is not None:
            scale = self.allreduce(self.flat.grads)
        self.flat.step += 1
        eng.adam(self.flat.params, self.flat.grads, self.flat.exp_avg, self.flat.exp_avg_sq,
                 self.flat.step, self.lr, grad_scale=scale)
        self.pack()
        return out, scalars
