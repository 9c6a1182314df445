"""Input pipeline of the training driver: image folders are decoded on the
host to **uint8** ``[C, H, W]`` tensors (RGB forced, like the reference's ``my_pil_loader``,
data/datasets.py:17-22) and uploaded as uint8; ``ToTensor`` (x / 255) runs inside the HIP
gather kernel.  A synthetic dataset of the same interface serves benchmarks and smoke runs
(no dataset can be downloaded here)."""

import os
from typing import Dict, List, Tuple

import torch as th
from torch.utils.data import Dataset


class SyntheticImages(Dataset):
    """Uniform random uint8 images with random labels: shapes only."""

    def __init__(self, n: int, channels: int, size: int, nb_class: int, seed: int = 0) -> None:
        g = th.Generator().manual_seed(seed)
        self.x = th.randint(0, 256, (n, channels, size, size), dtype=th.uint8, generator=g)
        self.y = th.randint(0, nb_class, (n,), generator=g)
        self.class_to_idx: Dict[str, int] = {str(i): i for i in range(nb_class)}

    def __len__(self) -> int:
        return self.x.shape[0]

    def __getitem__(self, i: int) -> Tuple[th.Tensor, th.Tensor]:
        return self.x[i], self.y[i]


class ImageFolderU8(Dataset):
    """``root/<class>/<image>`` -> (uint8 [3, H, W], label); every image must have the
    configured side (the reference assumes square, equally sized images)."""

    EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")

    def __init__(self, root: str, img_size: int = 0) -> None:
        from PIL import Image  # host-side decode only

        self._Image = Image
        self._img_size = img_size  # > 0: every image must be img_size x img_size
        classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        self.class_to_idx = {c: i for i, c in enumerate(classes)}
        self.items: List[Tuple[str, int]] = []
        for c in classes:
            for f in sorted(os.listdir(os.path.join(root, c))):
                if f.lower().endswith(self.EXT):
                    self.items.append((os.path.join(root, c, f), self.class_to_idx[c]))

    def __len__(self) -> int:
        return len(self.items)

    def __getitem__(self, i: int) -> Tuple[th.Tensor, th.Tensor]:
        import numpy as np

        path, label = self.items[i]
        with open(path, "rb") as f:
            img = self._Image.open(f).convert("RGB")
        arr = th.from_numpy(np.asarray(img).copy())  # [H, W, 3] uint8
        if self._img_size and (arr.shape[0] != self._img_size or arr.shape[1] != self._img_size):
            raise ValueError(f"{path}: {arr.shape[1]}x{arr.shape[0]} image, --img-size is {self._img_size}")
        return arr.permute(2, 0, 1).contiguous(), th.tensor(label)


class DevicePrefetcher:
    """Double-buffered host -> device upload.  The reference moves every batch
    synchronously at the top of the iteration (training/trainer.py:67-68) behind a pin_memory
    DataLoader (train.py:91-107); here batch i+1 is copied (uint8: a quarter of the fp32 bytes,
    ToTensor runs in the gather kernel) from pinned staging memory on a COPY stream while batch
    i trains, and the compute stream only waits for the copy's event."""

    def __init__(self, loader, device: th.device) -> None:
        self.loader = loader
        self.device = th.device(device)

    def _upload(self, batch, stream):
        if batch is None:
            return None
        x, y = batch
        if not x.is_pinned():
            x = x.pin_memory()
        if not y.is_pinned():
            y = y.pin_memory()
        with th.cuda.stream(stream):
            xd = x.to(self.device, non_blocking=True)
            yd = y.to(self.device, non_blocking=True)
        ev = th.cuda.Event()
        ev.record(stream)
        return xd, yd, ev, (x, y)  # keep the pinned staging tensors alive until the copy ran

    def __iter__(self):
        copy_stream = th.cuda.Stream(device=self.device)
        it = iter(self.loader)
        nxt = self._upload(next(it, None), copy_stream)
        while nxt is not None:
            xd, yd, ev, _host = nxt
            nxt = self._upload(next(it, None), copy_stream)  # overlaps with the step below
            cur = th.cuda.current_stream(self.device)
            cur.wait_event(ev)
            xd.record_stream(cur)
            yd.record_stream(cur)
            yield xd, yd

    def __len__(self) -> int:
        return len(self.loader)


class _Stripe:
    """every k-th batch of a batch sampler (same order: the base sampler is re-iterated per stripe)"""

    def __init__(self, base, k: int, j: int) -> None:
        self.base, self.k, self.j = base, k, j

    def __iter__(self):
        for i, b in enumerate(self.base):
            if i % self.k == self.j:
                yield b

    def __len__(self) -> int:
        n = len(self.base)
        return (n - self.j + self.k - 1) // self.k if n > self.j else 0


class StripedLoader:
    """``stripes`` DataLoaders over one batch sampler, batch i from loader i % stripes, yielded in order.
    One DataLoader funnels every batch through ONE collate / pin-memory thread of the parent process:
    measured 12.4 k images/s (256 x 256 uint8) whether 6, 32 or 96 workers decode; k stripes have k such
    threads (profiles/r04_loader.json)."""

    def __init__(self, dataset, batch_sampler, workers: int, stripes: int, **kw) -> None:
        from torch.utils.data import DataLoader

        self.batch_sampler = batch_sampler
        stripes = max(1, min(stripes, max(1, workers)))
        per = max(1, workers // stripes) if workers > 0 else 0
        self.loaders = [DataLoader(dataset, batch_sampler=_Stripe(batch_sampler, stripes, j), num_workers=per,
                                   pin_memory=True, persistent_workers=per > 0,
                                   prefetch_factor=4 if per > 0 else None, **kw) for j in range(stripes)]

    def __iter__(self):
        its = [iter(dl) for dl in self.loaders]
        i = 0
        while True:
            b = next(its[i % len(its)], None)
            if b is None:
                return
            yield b
            i += 1

    def __len__(self) -> int:
        return len(self.batch_sampler)


class ResidentLoader:
    """The whole image set decoded ONCE and kept in HBM as uint8 (``[N, C, H, W]``); every
    later batch is a device-side row gather, so from the second pass on the input pipeline
    costs no host work at all.  The reference re-decodes every image every epoch behind 6
    DataLoader workers (train.py:91-107), which one MI355X out-runs (input
    pipeline); 288 GB of HBM hold e.g. all of AID (10 000 x 3 x 600 x 600 B = 10.8 GB) many
    times over.  With ``world > 1`` each rank decodes ``1/world`` of the images and the
    shards are exchanged with one all-gather, so the host work is not repeated per rank.

    ``batch_sampler`` yields lists of dataset indices (this rank's slice of every global
    batch, train.ShardedBatchSampler); ``indices`` is the set it draws from."""

    def __init__(self, dataset, indices, batch_sampler, device, workers: int = 0,
                 rank: int = 0, world: int = 1, group=None, chunk: int = 16) -> None:
        self.dataset, self.batch_sampler = dataset, batch_sampler
        self.indices = list(indices)
        self.device = th.device(device)
        self.workers, self.rank, self.world, self.group, self.chunk = workers, rank, world, group, chunk
        self._x = self._y = self._row_of = None
        self.fill_img_s = float("nan")

    @staticmethod
    def nbytes(dataset, n: int) -> int:
        x, _ = dataset[0]
        return n * x.numel() * x.element_size()

    def _fill(self) -> None:
        from torch.utils.data import DataLoader

        n = len(self.indices)
        n_loc = -(-n // self.world)
        mine = self.indices[self.rank::self.world]
        mine = mine + [self.indices[-1]] * (n_loc - len(mine))  # equal shards for the all-gather
        chunks = [mine[i: i + self.chunk] for i in range(0, n_loc, self.chunk)]
        pin = self.device.type == "cuda"
        dl = DataLoader(self.dataset, batch_sampler=chunks, num_workers=self.workers, pin_memory=pin,
                        prefetch_factor=4 if self.workers > 0 else None)
        x_loc = y_loc = None
        at = 0
        import time

        t_first = None  # (fill_img_s: rate from the first chunk on - worker start-up excluded)
        for x, y in dl:
            if t_first is None:
                t_first, n_first = time.perf_counter(), x.shape[0]
            if x_loc is None:
                x_loc = th.empty((n_loc,) + tuple(x.shape[1:]), dtype=x.dtype, device=self.device)
                y_loc = th.empty((n_loc,), dtype=y.dtype, device=self.device)
            x_loc[at: at + x.shape[0]].copy_(x, non_blocking=pin)
            y_loc[at: at + x.shape[0]].copy_(y, non_blocking=pin)
            at += x.shape[0]
        assert at == n_loc
        if self.device.type == "cuda":
            th.cuda.synchronize(self.device)
        dt = time.perf_counter() - t_first if t_first is not None else 0.0
        self.fill_img_s = (n_loc - n_first) / dt if dt > 0 and n_loc > n_first else float("nan")
        if self.world > 1:
            import torch.distributed as dist

            # ONE [world * n_loc, ...] buffer filled in place: peak = (1 + 1 / world) x the image set
            x_all = th.empty((self.world * n_loc,) + tuple(x_loc.shape[1:]), dtype=x_loc.dtype, device=self.device)
            y_all = th.empty((self.world * n_loc,), dtype=y_loc.dtype, device=self.device)
            dist.all_gather_into_tensor(x_all, x_loc, group=self.group)
            dist.all_gather_into_tensor(y_all, y_loc, group=self.group)
            x_loc, y_loc = x_all, y_all
        # indices[j] was decoded by rank j % world as its (j // world)-th image
        j = th.arange(n)
        row_of = th.full((max(self.indices) + 1,), -1, dtype=th.long)
        row_of[th.tensor(self.indices, dtype=th.long)] = (j % self.world) * n_loc + j // self.world
        self._x, self._y, self._row_of = x_loc, y_loc, row_of.to(self.device)

    def __iter__(self):
        if self._x is None:
            self._fill()
        for batch in self.batch_sampler:
            rows = self._row_of[th.tensor(batch, dtype=th.long).to(self.device, non_blocking=True)]
            yield self._x.index_select(0, rows), self._y.index_select(0, rows)

    def __len__(self) -> int:
        return len(self.batch_sampler)
