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
