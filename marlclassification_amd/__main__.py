"""``python -m marlclassification_amd`` - the reference's command line (__main__.py:19-417):
``[-a N --step T --cuda --run-id ID] train [options]``, same flags, defaults and action
syntax (``[[1,0],[-1,0],...]``), with the ``train``, ``test`` and ``infer`` modes."""

import argparse
import re
from os.path import abspath, dirname, join

from .config import EvalConfig, InferConfig, MainConfig, ModelConfig, TrainConfig
from .networks.vision import CNN_BY_NAME

_TRAIN_FLAGS = (
    # flags, dest, type, default
    (("--action",), "action", str, "[[1, 0], [-1, 0], [0, 1], [0, -1]]"),
    (("--img-size",), "img_size", int, 28),
    (("--nb-class",), "nb_class", int, 10),
    (("-d", "--dim"), "dim", int, 2),
    (("--f",), "f", int, 7),
    (("--nb",), "n_b", int, 64),
    (("--na",), "n_a", int, 16),
    (("--nm",), "n_m", int, 16),
    (("--nmo",), "n_m_o", int, 24),
    (("--nd",), "n_d", int, 4),
    (("--nlb",), "n_l_b", int, 128),
    (("--nla",), "n_l_a", int, 128),
    (("--batch-size",), "batch_size", int, 8),
    (("--lr", "--learning-rate"), "learning_rate", float, 1e-3),
    (("--gamma",), "gamma", float, 0.99),
    (("--nb-epoch",), "nb_epoch", int, 10),
)


def parse_actions(text: str, dim: int):
    compact = text.replace(" ", "")
    if not re.match(r"^\[(\[(-?\d+,?)+\],)*\[(-?\d+,?)+\]\]$", compact):
        raise ValueError(f"Wrong action(s) : {text}")
    actions = [[int(v) for v in a.split(",")] for a in re.findall(r"\[((?:-?\d+,?)+)\]", compact)]
    for i, a in enumerate(actions):
        assert len(a) == dim, f"Wrong space for action at index {i}"
    return actions


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser("python -m marlclassification_amd")
    p.add_argument("--run-id", type=str, required=True, dest="run_id")
    p.add_argument("-a", "--agents", type=int, default=3, dest="agents")
    p.add_argument("--step", type=int, default=7)
    p.add_argument("--cuda", action="store_true", dest="cuda")
    sub = p.add_subparsers(dest="main_choice", required=True)
    t = sub.add_parser("train")
    for flags, dest, typ, default in _TRAIN_FLAGS:
        t.add_argument(*flags, type=typ, default=default, dest=dest)
    t.add_argument("--ft-extr", type=str, choices=sorted(CNN_BY_NAME), default="mnist", dest="ft_extr_str")
    t.add_argument("--res-folder", type=str, dest="res_folder",
                   default=abspath(join(dirname(abspath(__file__)), "..", "resources")))
    t.add_argument("-o", "--output-dir", type=str, required=True, dest="output_dir")
    e = sub.add_parser("test")  # reference __main__.py:217-258
    e.add_argument("--batch-size", type=int, default=8, dest="batch_size")
    e.add_argument("--dataset-path", type=str, required=True, dest="dataset_path")
    e.add_argument("--img-size", type=int, default=28, dest="img_size")
    e.add_argument("--json-path", type=str, required=True, dest="json_path")
    e.add_argument("--state-dict-path", type=str, required=True, dest="state_dict_path")
    e.add_argument("-o", "--output-dir", type=str, required=True, dest="output_dir")
    i = sub.add_parser("infer")  # reference __main__.py:263-300
    i.add_argument("--images", type=str, nargs="+", required=True, dest="infer_images")
    i.add_argument("--json-path", type=str, required=True, dest="json_path")
    i.add_argument("--state-dict-path", type=str, required=True, dest="state_dict_path")
    i.add_argument("--class2idx", type=str, required=True, dest="class_to_idx")
    i.add_argument("-o", "--output-image-dir", type=str, required=True, dest="output_image_dir")
    t.add_argument("--exact-standardize", action="store_true", dest="exact_standardize",
                   help="multi-GPU: global advantage statistics (update == single-GPU big batch)")
    return p


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    main_config = MainConfig(step=args.step, run_id=args.run_id, cuda=args.cuda, nb_agent=args.agents)
    if args.main_choice == "train":
        from .train import train_main

        model_config = ModelConfig(
            ft_extr_str=args.ft_extr_str, window_size=args.f, hidden_size_belief=args.n_b,
            hidden_size_action=args.n_a, hidden_size_msg=args.n_m,
            hidden_size_msg_output=args.n_m_o, hidden_size_state=args.n_d, state_dim=args.dim,
            actions=parse_actions(args.action, args.dim), nb_class=args.nb_class,
            hidden_size_linear_belief=args.n_l_b, hidden_size_linear_action=args.n_l_a,
        )
        train_config = TrainConfig(
            img_size=args.img_size, nb_epoch=args.nb_epoch, learning_rate=args.learning_rate,
            batch_size=args.batch_size, resources_dir=args.res_folder, output_dir=args.output_dir,
            gamma=args.gamma,
        )
        train_main(main_config, model_config, train_config, exact_standardize=args.exact_standardize)
    elif args.main_choice == "test":
        from .eval import eval_main

        eval_main(main_config, EvalConfig(
            img_size=args.img_size, state_dict_path=args.state_dict_path, batch_size=args.batch_size,
            json_path=args.json_path, dataset_path=args.dataset_path, output_dir=args.output_dir))
    elif args.main_choice == "infer":
        import os

        from .infer import infer_main

        if os.path.exists(args.output_image_dir) and not os.path.isdir(args.output_image_dir):
            raise NotADirectoryError(f'"{args.output_image_dir}" is not a directory.')
        os.makedirs(args.output_image_dir, exist_ok=True)
        infer_main(main_config, InferConfig(
            state_dict_path=args.state_dict_path, json_path=args.json_path, images_path=args.infer_images,
            output_dir=args.output_image_dir, class_to_idx=args.class_to_idx))


if __name__ == "__main__":
    main()
