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

    def __init__(self, root: str) -> None:
        from PIL import Image  # host-side decode only

        self._Image = Image
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
        return arr.permute(2, 0, 1).contiguous(), th.tensor(label)
