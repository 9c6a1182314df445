"""Trainer: the A2C loop of the reference (training/trainer.py) on the HIP path.

One training iteration is: ``marl_episode_forward`` (all steps) -> ``marl_a2c_loss_fwd_bwd``
(loss + dL/d outputs) -> ``marl_episode_backward`` (BPTT, every parameter gradient) ->
[one RCCL all-reduce of the flat gradient buffer] -> ``marl_adam_step`` -> re-pack.  No host
synchronisation happens inside an iteration; meters are read every ``log_interval`` steps.
Same constructor and ``train_epoch`` / ``eval_epoch`` signatures as the reference.
"""

from typing import Callable, Dict, Iterable, Optional, Tuple

import torch as th

from ..core import EpisodeSampler
from ..engine import EpisodeTensors
from ..metrics import ConfusionMeter, LossMeter
from ..networks import ModelsWrapper

MetricLogger = Callable[[int, Dict[str, float]], None]


class Trainer:
    def __init__(
        self,
        model: ModelsWrapper,
        nb_class: int,
        learning_rate: float,
        gamma: float,
        metric_logger: Optional[MetricLogger] = None,
        log_interval: int = 100,
        meter_window_size: int = 64,
        allreduce: Optional[Callable[[th.Tensor], float]] = None,
        exact_standardize_group=None,
    ) -> None:
        self.__model = model
        self.__nb_class = nb_class
        self.__lr = learning_rate
        self.__gamma = gamma
        self.__metric_logger = metric_logger
        self.__log_interval = log_interval
        self.__allreduce = allreduce
        self.__exact_group = exact_standardize_group
        self.__curr_step = 0
        self.__loss_bufs: Optional[Tuple[th.Tensor, ...]] = None
        self.__conf_meter = ConfusionMeter(nb_class, window_size=meter_window_size)
        self.__meters = {k: LossMeter(window_size=meter_window_size)
                         for k in ("loss", "path", "error", "critic")}

    @property
    def curr_step(self) -> int:
        return self.__curr_step

    # one optimisation step on a batch (reference trainer.py:67-116)
    def train_step(self, x: th.Tensor, y: th.Tensor, sampler: EpisodeSampler) -> Tuple[EpisodeTensors, th.Tensor]:
        model = self.__model
        device = model.device
        y = y.to(device)
        eng, out = sampler.run_episode_raw(x, train=True)
        if self.__loss_bufs is None or self.__loss_bufs[0].shape != out.step_preds.shape:
            self.__loss_bufs = (
                th.empty_like(out.step_preds), th.empty_like(out.step_log_probas),
                th.empty_like(out.step_values), th.zeros(4, device=device),
                th.zeros(3, dtype=th.float64, device=device),
            )
        if self.__exact_group is None:
            gp, gl, gv, scalars, _ = eng.a2c_loss(out, y, self.__gamma, 0, self.__loss_bufs)
        else:  # global mean / std of the advantages: one 3-double all-reduce between phases
            from ..parallel import allreduce_adv_stats

            _, _, _, _, stats = eng.a2c_loss(out, y, self.__gamma, 1, self.__loss_bufs)
            allreduce_adv_stats(stats, self.__exact_group)
            gp, gl, gv, scalars, _ = eng.a2c_loss(out, y, self.__gamma, 2, self.__loss_bufs)
        flat = model.flat_state()
        bucketed = hasattr(self.__allreduce, "before_backward")  # parallel.BucketedGradAllReduce
        if bucketed:
            self.__allreduce.before_backward(eng)
        try:
            eng.episode_backward(gp, gl, gv, flat.grad_views())
        finally:
            if bucketed:
                self.__allreduce.after_backward(eng)
        scale = 1.0 if self.__allreduce is None else self.__allreduce(flat.grads)
        flat.step += 1
        eng.adam(flat.params, flat.grads, flat.exp_avg, flat.exp_avg_sq, flat.step, self.__lr,
                 grad_scale=scale)
        model.mark_updated(eng)
        return out, scalars

    def train_epoch(self, dataloader: Iterable, epoch_index: int, episode_sampler: EpisodeSampler) -> None:
        self.__model.train()
        for x_train, y_train in dataloader:
            out, scalars = self.train_step(x_train, y_train, episode_sampler)
            # device-side meters, no sync (select last step, mean over agents: trainer.py:124-128)
            self.__conf_meter.add(out.step_preds[-1].mean(dim=0), y_train)
            for i, k in enumerate(("loss", "path", "error", "critic")):
                self.__meters[k].add(scalars[i].clone())
            if self.__metric_logger is not None and self.__curr_step % self.__log_interval == 0:
                self.__metric_logger(self.__curr_step, self.metrics())
            self.__curr_step += 1

    def metrics(self) -> Dict[str, float]:
        """Synchronises: windowed means of the loss terms + train precision / recall."""
        return {
            "error": self.__meters["error"].loss(),
            "path_loss": self.__meters["path"].loss(),
            "loss": self.__meters["loss"].loss(),
            "critic_loss": self.__meters["critic"].loss(),
            "train_prec": self.__conf_meter.precision().mean().item(),
            "train_rec": self.__conf_meter.recall().mean().item(),
        }

    def eval_epoch(self, dataloader: Iterable, epoch_index: int, episode_sampler: EpisodeSampler) -> ConfusionMeter:
        self.__model.eval()
        conf_meter = ConfusionMeter(self.__nb_class, None)
        with th.no_grad():
            for x_test, y_test in dataloader:
                out = episode_sampler.run_episode_get_last_step(x_test)
                conf_meter.add(out.prediction.mean(dim=0), y_test)
        return conf_meter
