"""``train_main``: the reference's training driver (train.py:18-168) on the HIP path -
same configs, same output tree (``marl.json``, ``class_to_idx.json``,
``models/nn_models_epoch_{e}.pt`` with reference state-dict keys).  MLflow is optional (not
installed here); images are uploaded as uint8 (data.py).  Multi-GPU: launch with
``torchrun --nproc-per-node N -m marlclassification_amd ...``; every rank trains on its
shard of each batch and gradients are all-reduced over RCCL."""

import json
import os
from os.path import exists, isdir, join
from typing import Dict, Optional

import torch as th
from torch.utils.data import DataLoader, Sampler

from .config import MainConfig, ModelConfig, TrainConfig
from .core import EpisodeSampler
from .data import DevicePrefetcher, ImageFolderU8, ResidentLoader, StripedLoader, SyntheticImages
from .parallel import BucketedGradAllReduce
from .training import Trainer


# where each image-folder dataset lives under the resources directory (reference
# data/datasets.py:25-79,217-235: MnistDataset / Resisc45Dataset / AidDataset / SkinCancerDataset)
_DATASET_SUBDIR = {
    "mnist": ("downloaded", "mnist_png", "all_png"),
    "resisc45": ("downloaded", "NWPU-RESISC45"),
    "aid": ("downloaded", "AID"),
    "skin_cancer": ("downloaded", "skin_cancer"),
}


def _dataset(model_config: ModelConfig, train_config: TrainConfig):
    """``--res-folder synthetic`` (explicit) = random images of the configured shape; anything else
    must be the reference's resources directory: the dataset is read from the same sub-folder
    the reference's dataset class uses, and a missing folder is an error (never silent noise)."""
    root = train_config.resources_dir
    if root == "synthetic":
        channels = 1 if model_config.ft_extr_str == "mnist" else 3
        return SyntheticImages(max(4 * train_config.batch_size, 64), channels, train_config.img_size,
                               model_config.nb_class)
    sub = _DATASET_SUBDIR.get(model_config.ft_extr_str)
    if sub is None:
        raise ValueError(f'no image-folder dataset is known for "{model_config.ft_extr_str}" '
                         f"(supported: {sorted(_DATASET_SUBDIR)}, or --res-folder synthetic)")
    path = join(root, *sub)
    if not (exists(path) and isdir(path)):
        raise NotADirectoryError(f'"{path}" does not exist or is not a directory '
                                 "(pass the reference's resources folder, or --res-folder synthetic)")
    dataset = ImageFolderU8(path, img_size=train_config.img_size)
    if len(dataset) == 0 or len(dataset.class_to_idx) != model_config.nb_class:
        raise ValueError(f'"{path}": {len(dataset)} images in {len(dataset.class_to_idx)} classes, '
                         f"--nb-class is {model_config.nb_class}")
    return dataset


class ShardedBatchSampler(Sampler):
    """Batches of dataset INDICES for one rank.  Every rank derives the same shuffled order of
    the same index list from (seed, epoch) and takes its contiguous slice of every global batch,
    so a rank only decodes the images it trains on (the reference has one process and 6 loader
    workers, train.py:91-107; slicing after the decode would repeat the host work on every
    rank).  The last, smaller global batch is trimmed to a multiple of the world size."""

    def __init__(self, indices, global_batch: int, rank: int, world: int, shuffle: bool, seed: int) -> None:
        if global_batch % world != 0:
            raise ValueError(f"batch size {global_batch} is not divisible by the world size {world}")
        self.indices = list(indices)
        self.global_batch, self.rank, self.world = global_batch, rank, world
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _batches(self):
        order = self.indices
        if self.shuffle:
            g = th.Generator().manual_seed(self.seed * 1_000_003 + self.epoch)
            order = [self.indices[i] for i in th.randperm(len(order), generator=g).tolist()]
        for lo in range(0, len(order), self.global_batch):
            chunk = order[lo: lo + self.global_batch]
            n = (len(chunk) // self.world) * self.world
            if n == 0:
                continue
            per = n // self.world
            yield chunk[self.rank * per: (self.rank + 1) * per]

    def __iter__(self):
        return self._batches()

    def __len__(self) -> int:
        full, rest = divmod(len(self.indices), self.global_batch)
        return full + (1 if rest >= self.world else 0)


def loader_workers(dataset) -> int:
    """Decode processes per rank: the reference uses 6 (train.py:95); MARL_LOADER_WORKERS
    overrides; in-memory synthetic data needs none."""
    if isinstance(dataset, SyntheticImages):
        return 0
    env = os.environ.get("MARL_LOADER_WORKERS")
    if env is not None:
        return max(0, int(env))
    return max(0, min(6, (os.cpu_count() or 1) - 1))


def resident_fits(dataset) -> bool:
    """Keep the decoded uint8 image set in HBM (data.ResidentLoader) when it is an image
    folder no larger than MARL_RESIDENT_GB (default 64; 0 = always stream from the loader)."""
    if not isinstance(dataset, ImageFolderU8) or len(dataset) == 0:
        return False
    budget = float(os.environ.get("MARL_RESIDENT_GB", "64")) * 1e9
    # peak while the shards are exchanged: this rank's 1/world + the gathered whole (data.ResidentLoader._fill)
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    return ResidentLoader.nbytes(dataset, len(dataset)) * (1.0 + (1.0 / world if world > 1 else 0.0)) <= budget


def train_main(main_config: MainConfig, model_config: ModelConfig, train_config: TrainConfig,
               metric_logger=None, exact_standardize: bool = False) -> Trainer:
    """``exact_standardize`` (multi-GPU): one extra 3-double all-reduce per iteration makes the
    advantage standardisation global, so the update equals the single-GPU big-batch update."""
    assert model_config.state_dim == 2, "the HIP path implements 2-D images (state_dim == 2)"
    output_dir = train_config.output_dir
    model_dir = join(output_dir, "models")
    os.makedirs(model_dir, exist_ok=True)
    if not isdir(model_dir):
        raise NotADirectoryError(f'"{model_dir}" is not a directory.')
    if not main_config.cuda:
        raise RuntimeError("this implementation only runs on the GPU: pass --cuda")

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (any world size: one rank still goes through RCCL, so
    # the exact command line of a multi-GPU run is what a 1-GPU box tests)
    distributed = "RANK" in os.environ
    device = th.device("cuda", local_rank)
    th.cuda.set_device(device)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=device)
    # one base seed for the run: identical initial weights and data order on every rank; the episode draws
    # mix the rank in (core/episode.py), so shards draw differently.  MARL_SEED unset: a fresh seed per run
    # like the reference (train.py never seeds), agreed on by all ranks; it is printed, so a run can be
    # repeated with MARL_SEED=<that value>.
    if "MARL_SEED" in os.environ:
        base_seed = int(os.environ["MARL_SEED"])
    else:
        t = th.tensor([int.from_bytes(os.urandom(4), "little")], dtype=th.int64, device=device)
        if distributed:
            dist.broadcast(t, 0)
        base_seed = int(t.item())
    th.manual_seed(base_seed)
    if rank == 0:
        print(f"seed {base_seed}, world size {world}", flush=True)

    nn_models, marl_m, env = model_config.build_marl(main_config.nb_agent)
    dataset = _dataset(model_config, train_config)
    if rank == 0:
        model_config.save_marl_config(join(output_dir, "marl.json"))
        with open(join(output_dir, "class_to_idx.json"), "w", encoding="utf-8") as f:
            json.dump(dataset.class_to_idx, f)
    nn_models.to(device)
    if distributed:  # identical initial weights on every rank
        for p in nn_models.parameters():
            dist.broadcast(p.data, src=0)

    g = th.Generator().manual_seed(base_seed)  # same split on every rank
    idx = th.randperm(len(dataset), generator=g)
    cut = int(0.85 * idx.shape[0])
    loaders, samplers = [], []
    workers = loader_workers(dataset)
    resident = resident_fits(dataset)
    if rank == 0:
        print(f"input pipeline: {'HBM-resident uint8 image set' if resident else 'streamed'}, "
              f"{workers} decode processes per rank", flush=True)
    for k, part in enumerate((idx[:cut].tolist(), idx[cut:].tolist())):
        bs = ShardedBatchSampler(part, train_config.batch_size, rank, world, shuffle=True,
                                 seed=base_seed + 1 + k)
        samplers.append(bs)
        if resident:  # decoded once (1/world per rank), then batches are row gathers in HBM
            loaders.append(ResidentLoader(dataset, part, bs, device, workers=workers, rank=rank, world=world))
            continue
        # (MARL_LOADER_STRIPES DataLoaders share the workers: one collate / pin thread each, data.StripedLoader)
        dl = StripedLoader(dataset, bs, workers, int(os.environ.get("MARL_LOADER_STRIPES", max(1, workers // 8))))
        loaders.append(DevicePrefetcher(dl, device))  # upload of batch i+1 overlaps step i

    sampler = EpisodeSampler(marl_m, env, main_config.step)
    trainer = Trainer(nn_models, marl_m.nb_class, train_config.learning_rate, train_config.gamma,
                      metric_logger=metric_logger if rank == 0 else None,
                      allreduce=(BucketedGradAllReduce(world, None, nn_models.flat_state().offsets, nn_models.flat_state().numel,
                                                      device) if distributed else None),
                      exact_standardize_group=(dist.group.WORLD if distributed and world > 1 and
                                               exact_standardize else None))
    for e in range(train_config.nb_epoch):
        for bs in samplers:
            bs.set_epoch(e)
        trainer.train_epoch(loaders[0], e, sampler)
        conf = trainer.eval_epoch(loaders[1], e, sampler)
        if distributed:
            conf.all_reduce()  # every rank evaluated its shard only
        if rank == 0:
            m: Dict[str, float] = trainer.metrics()
            m["eval_prec"] = conf.precision().mean().item()
            m["eval_recs"] = conf.recall().mean().item()
            print(f"epoch {e}: " + ", ".join(f"{k}={v:.4f}" for k, v in m.items()), flush=True)
            conf.save_conf_matrix(e, output_dir, "eval")  # reference train.py:134
            th.save(nn_models.state_dict(), join(model_dir, f"nn_models_epoch_{e}.pt"))
    if rank == 0 and len(idx) > cut:
        # one evaluation image, step by step (reference train.py:149-166): frames + GIF in output_dir
        from .visualization import visualize_steps

        pick = int(idx[cut + int(th.randint(0, len(idx) - cut, (1,)).item())])
        x_u8 = dataset[pick][0]
        visualize_steps(sampler, x_u8.to(device), x_u8, model_config.window_size, output_dir,
                        dataset.class_to_idx)
    if distributed:
        dist.destroy_process_group()
    return trainer
