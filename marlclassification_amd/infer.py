"""``infer_main``: the reference's ``infer`` mode (infer.py:17-95) on the HIP path - for every image
matched by the given glob patterns, one episode with the loaded model and the step-by-step
visualisation (PNG frames + GIF) in ``<output_dir>/<image name>/``."""

import glob
import json
import os
from os.path import exists, isfile, join, split

import torch as th

from .config import InferConfig, MainConfig, ModelConfig
from .core import EpisodeSampler
from .visualization import visualize_steps


def load_image_u8(path: str) -> th.Tensor:
    """RGB uint8 [3, H, W] (the reference's ``my_pil_loader`` forces RGB, data/datasets.py:17-22)."""
    import numpy as np
    from PIL import Image

    with open(path, "rb") as f:
        arr = np.asarray(Image.open(f).convert("RGB")).copy()
    return th.from_numpy(arr).permute(2, 0, 1).contiguous()


def infer_main(main_config: MainConfig, infer_config: InferConfig) -> int:
    for what, path in (("JSON path", infer_config.json_path), ("State dict path", infer_config.state_dict_path),
                       ("class_to_idx", infer_config.class_to_idx)):
        assert exists(path), f'{what} "{path}" does not exist'
        assert isfile(path), f'"{path}" is not a file'
    if not main_config.cuda:
        raise RuntimeError("this implementation only runs on the GPU: pass --cuda")
    device = th.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    with open(infer_config.class_to_idx, "r", encoding="utf-8") as f:
        class_to_idx = json.load(f)
    marl_config = ModelConfig.load_marl_config(infer_config.json_path)
    nn_models, marl_m, env = marl_config.build_marl(main_config.nb_agent)
    nn_models.load_state_dict(th.load(infer_config.state_dict_path, map_location="cpu"))
    nn_models.eval()
    nn_models.to(device)
    sampler = EpisodeSampler(marl_m, env, main_config.step)

    paths = sorted(p for pattern in infer_config.images_path for p in glob.glob(pattern, recursive=True))
    for img_path in paths:
        x = load_image_u8(img_path)
        out_dir = join(infer_config.output_dir, split(img_path)[-1])
        os.makedirs(out_dir, exist_ok=True)
        with open(join(out_dir, "info.txt"), "w", encoding="utf-8") as info:
            info.write(f"{img_path}\n{infer_config.json_path}\n{infer_config.state_dict_path}\n")
        visualize_steps(sampler, x.to(device), x, marl_config.window_size, out_dir, class_to_idx)
    return len(paths)
