"""``visualize_steps`` (reference visualization.py:12-99): runs ONE episode on one image on the HIP
path and paints, step after step, the windows the agents have visited, with the running class
vote in the title; writes ``pred_original.png``, ``pred_step_{t}.png`` and ``animated_gif.gif``.
Consumer of the ``step_pos`` / ``step_preds`` layout of ``EpisodeSampler.run_episode``; host-side
reporting (matplotlib / PIL), nothing here is on the timed path."""

from os.path import join
from typing import Any, List, Mapping

import torch as th

from .core import EpisodeSampler


def visualize_steps(episode_sampler: EpisodeSampler, img: th.Tensor, img_ori: th.Tensor,
                    window_size: int, output_dir: str, class_map: Mapping[Any, int]) -> List[str]:
    """``img`` [C,H,W] goes through the episode (fp32 in [0,1] or uint8); ``img_ori`` [C,H,W] is what
    gets painted.  Returns the written file names."""
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from PIL import Image

    names = {idx: name for name, idx in class_map.items()}
    with th.no_grad():
        out = episode_sampler.run_episode(img.unsqueeze(0))
    votes = th.softmax(out.step_preds.float().mean(dim=1)[:, 0].cpu(), dim=-1)  # [Ns, nC]
    pos = out.step_pos[:, :, 0].cpu()                                            # [Ns, Na, 2]
    canvas_src = img_ori.detach().cpu()
    if canvas_src.dtype == th.uint8:
        canvas_src = canvas_src.float() / 255.0
    canvas_src = canvas_src.permute(1, 2, 0)
    if canvas_src.shape[2] == 1:
        canvas_src = canvas_src.repeat(1, 1, 3)
    h, w, _ = canvas_src.shape

    written: List[str] = []

    def frame(array, title: str, name: str) -> str:
        fig = plt.figure()
        plt.imshow(array)
        plt.title(title)
        path = join(output_dir, name)
        plt.savefig(path)
        plt.close(fig)
        written.append(path)
        return path

    first = frame(canvas_src.numpy(), "Original", "pred_original.png")
    frames = [Image.open(first) for _ in range(5)]  # 5 x 200 ms: the original stays for a second
    seen = th.zeros(h, w, 4)  # RGBA: alpha marks what the agents have looked at so far
    for t in range(pos.shape[0]):
        for a in range(pos.shape[1]):
            r0, c0 = int(pos[t, a, 0]), int(pos[t, a, 1])
            seen[r0:r0 + window_size, c0:c0 + window_size, :3] = canvas_src[r0:r0 + window_size, c0:c0 + window_size]
            seen[r0:r0 + window_size, c0:c0 + window_size, 3] = 1.0
        best = int(votes[t].argmax())
        path = frame(seen.numpy(), f"Step = {t}, step_pred_class = {names[best]} "
                                   f"({votes[t, best].item() * 100.0:.1f}%)", f"pred_step_{t}.png")
        frames.append(Image.open(path))
    gif = join(output_dir, "animated_gif.gif")
    frames[0].save(gif, save_all=True, append_images=frames[1:], duration=200, loop=0)
    written.append(gif)
    return written
