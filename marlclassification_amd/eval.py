"""``eval_main``: the reference's ``test`` mode (eval.py:15-87) on the HIP path - load ``marl.json``
and a state dict (reference checkpoints load as they are), run every image of an image-folder
dataset through ``run_episode_get_last_step`` and report the confusion matrix, per-class
precision and recall.  Images go up as uint8 (ToTensor inside the gather kernel)."""

import os
from os.path import exists, isdir, isfile

import torch as th
from torch.utils.data import DataLoader

from .config import EvalConfig, MainConfig, ModelConfig
from .core import EpisodeSampler
from .data import DevicePrefetcher, ImageFolderU8
from .metrics import ConfusionMeter, format_metric


def eval_main(main_config: MainConfig, eval_config: EvalConfig) -> ConfusionMeter:
    for what, path in (("JSON path", eval_config.json_path), ("State dict path", eval_config.state_dict_path)):
        assert exists(path), f'{what} "{path}" does not exist'
        assert isfile(path), f'"{path}" is not a file'
    if exists(eval_config.output_dir) and not isdir(eval_config.output_dir):
        raise NotADirectoryError(f'"{eval_config.output_dir}" is not a directory')
    os.makedirs(eval_config.output_dir, exist_ok=True)
    if not main_config.cuda:
        raise RuntimeError("this implementation only runs on the GPU: pass --cuda")
    device = th.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))

    dataset = ImageFolderU8(eval_config.dataset_path, img_size=eval_config.img_size)
    marl_config = ModelConfig.load_marl_config(eval_config.json_path)
    nn_models, marl_m, env = marl_config.build_marl(main_config.nb_agent)
    nn_models.load_state_dict(th.load(eval_config.state_dict_path, map_location="cpu"))
    nn_models.eval()
    nn_models.to(device)
    loader = DevicePrefetcher(DataLoader(dataset, batch_size=eval_config.batch_size, shuffle=True,
                                         num_workers=0, drop_last=False, pin_memory=True), device)
    sampler = EpisodeSampler(marl_m, env, main_config.step)
    meter = ConfusionMeter(nn_models.nb_class)
    with th.no_grad():
        for x, y in loader:
            out = sampler.run_episode_get_last_step(x)
            meter.add(out.prediction.mean(dim=0), y)  # mean over agents; stays on the device

    print(meter.conf_mat().cpu())
    precs, recs = meter.precision(), meter.recall()
    print(f"Precision : {format_metric(precs, dataset.class_to_idx)}")
    print(f"Precision mean = {precs.mean().item()}")
    print(f"Recall : {format_metric(recs, dataset.class_to_idx)}")
    print(f"Recall mean : {recs.mean().item()}")
    meter.save_conf_matrix(0, eval_config.output_dir, "test")
    return meter
