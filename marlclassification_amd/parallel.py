"""Data parallelism over the GPUs of one node: one process per GPU, the batch sharded over
ranks (every image keeps all of its agents on one GPU - the only cross-row operation,
the message mean, reduces over agents of the SAME image: networks/message.py:17), weights
replicated, and ONE all-reduce of the flat fp32 gradient buffer per iteration
(``torch.distributed`` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code: this is new.  Equal shard sizes
make the mean over (agents, batch) of training/trainer.py:111 equal to the average of the
per-shard means, so averaging gradients reproduces the big-batch gradient, except for
``standardize`` (functions.py:54-55) whose statistics are per shard by default (what DDP
on the reference would do) and global with ``exact_standardize`` (one extra 3-double
all-reduce, C ABI phases 1/2 of marl_a2c_loss_fwd_bwd).
"""

from __future__ import annotations

import os
from typing import Tuple

import torch as th
import torch.distributed as dist


def shard_seed(base: int, rank: int) -> int:
    """Rank-offset seeds so that shards draw different positions / noise."""
    return base * 1_000_003 + 7919 * rank + 1


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Equal contiguous shards; the batch must divide evenly (keeps means exact)."""
    if n % world != 0:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return rank * per, (rank + 1) * per


def _all_reduce_sum(t: th.Tensor, group=None) -> th.Tensor:
    """Sum-all-reduce in place.  GPU tensors go straight to RCCL ("nccl" backend); under a
    "gloo" group (CPU-only rendezvous, the 2-rank tests that share one GPU) they bounce through
    the host."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class GradAllReduce:
    """Sum-all-reduce of the flat gradient buffer; returns the scale (1 / world) that the
    Adam kernel applies while it reads the gradient (no separate divide pass)."""

    def __init__(self, world: int, group=None) -> None:
        self.world = world
        self.group = group

    def __call__(self, flat_grads: th.Tensor) -> float:
        _all_reduce_sum(flat_grads, self.group)
        return 1.0 / self.world


#: parameters whose gradients are final before the reverse-time loop of the backward pass (csrc/episode.hip,
#: episode_backward: the batched heads) - the state-dict prefixes of Policy / Critic / Prediction
HEAD_PREFIXES = ("_ModelsWrapper__policy.", "_ModelsWrapper__critic.", "_ModelsWrapper__predict.")


class BucketedGradAllReduce(GradAllReduce):
    """Two buckets instead of one: the heads' slice of the flat gradient buffer - complete
    before the reverse loop starts - is all-reduced on a side stream while the loop, the batched weight gradients
    and the CNN backward still run; the rest follows on the main stream.  At 32 images per GPU (2.7 ms iterations)
    the single 6.7 MB exchange was otherwise fully exposed behind the backward pass.

    The heads must be ONE contiguous range of the flat buffer (they are the last three modules of the state dict:
    networks/models.py); otherwise this degrades to the single all-reduce.  Same sums element by element: at world
    size 2 the update is bit-equal to the one-bucket form (tested over gloo)."""

    def __init__(self, world: int, group, offsets: dict, numel: int, device: th.device) -> None:
        super().__init__(world, group)
        heads = sorted(off for name, off in offsets.items() if name.startswith(HEAD_PREFIXES))
        others = [off for name, off in offsets.items() if not name.startswith(HEAD_PREFIXES)]
        self.split = heads[0] if heads and (not others or max(others) < heads[0]) else None
        if os.environ.get("MARL_GRAD_BUCKETS") == "1":  # (escape hatch: the single all-reduce of rounds 1-5)
            self.split = None
        self.numel = numel
        self._event = None
        self._side = None
        if self.split is not None and device.type == "cuda":
            self._event = th.cuda.Event()
            self._event.record(th.cuda.current_stream(device))  # (creates the handle the library records later)
            self._side = th.cuda.Stream(device=device)
        self._armed = False

    def before_backward(self, engine) -> None:
        """Installs the event: the library records it where the heads' gradients are final."""
        if self.split is None:
            return
        if self._event is not None:
            engine.set_heads_event(self._event.cuda_event)
        self._armed = True

    def after_backward(self, engine) -> None:
        if self._event is not None:
            engine.set_heads_event(None)

    def __call__(self, flat_grads: th.Tensor) -> float:
        if not self._armed:
            return super().__call__(flat_grads)
        self._armed = False
        if self._side is None:  # (CPU tensors, the gloo tests: the same two collectives, one after the other)
            _all_reduce_sum(flat_grads[self.split:], self.group)
            _all_reduce_sum(flat_grads[: self.split], self.group)
            return 1.0 / self.world
        main = th.cuda.current_stream(flat_grads.device)
        self._side.wait_event(self._event)
        with th.cuda.stream(self._side):
            _all_reduce_sum(flat_grads[self.split:], self.group)
        _all_reduce_sum(flat_grads[: self.split], self.group)
        main.wait_stream(self._side)
        return 1.0 / self.world


def allreduce_adv_stats(stats: th.Tensor, group=None) -> th.Tensor:
    """(n, sum, sum of squares) of the advantages summed over ranks: the exchange step of
    the exact global ``standardize``."""
    return _all_reduce_sum(stats, group)


def broadcast_parameters(flat_params: th.Tensor, src: int = 0, group=None) -> None:
    """Identical initial weights on every rank: one broadcast of the flat parameter buffer."""
    if flat_params.is_cuda and dist.get_backend(group) == "gloo":
        host = flat_params.cpu()
        dist.broadcast(host, src=src, group=group)
        flat_params.copy_(host)
    else:
        dist.broadcast(flat_params, src=src, group=group)


def allreduce_confusion(conf_mat: th.Tensor, group=None) -> th.Tensor:
    """Evaluation metrics over the WHOLE evaluation set: sum the per-rank confusion matrices."""
    return _all_reduce_sum(conf_mat, group)
