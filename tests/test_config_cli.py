"""CPU: marl.json wire format and the command-line surface of the reference (SURVEY 8 f-2,
f-3) - field names, defaults, action syntax."""
import json

import pytest

from marlclassification_amd.__main__ import build_parser, parse_actions
from marlclassification_amd.config import ModelConfig

# resources/trained_models/mnist/marl.json of the reference, verbatim content (data fixture)
REF_MNIST_JSON = {
    "ft_extr_str": "mnist", "window_size": 6, "hidden_size_belief": 80, "hidden_size_action": 80,
    "hidden_size_msg": 16, "hidden_size_msg_output": 24, "hidden_size_state": 8, "state_dim": 2,
    "actions": [[1, 0], [-1, 0], [0, 1], [0, -1], [0, 0]], "nb_class": 10,
    "hidden_size_linear_belief": 112, "hidden_size_linear_action": 112,
}


def test_marl_json_roundtrip(tmp_path):
    p = tmp_path / "marl.json"
    p.write_text(json.dumps(REF_MNIST_JSON))
    cfg = ModelConfig.load_marl_config(str(p))
    assert cfg.actions[-1] == [0, 0] and cfg.hidden_size_belief == 80
    out = tmp_path / "out.json"
    cfg.save_marl_config(str(out))
    assert json.loads(out.read_text()) == REF_MNIST_JSON
    model, agents, env = cfg.build_marl(3)
    assert len(agents) == 3 and env.nb_actions == 5 and model.nb_class == 10
    assert sum(p.numel() for p in model.parameters()) == 149616  # the shipped checkpoint's size


def test_cli_matches_reference_readme_command():
    # README.md:41 of the reference (RESISC45 training command)
    argv = ("-a 16 --step 16 --cuda --run-id train_resisc45 train --action [[1,0],[-1,0],[0,1],[0,-1]] "
            "--ft-extr resisc45 --batch-size 8 --nb-class 45 --img-size 256 -d 2 --nb 256 --na 256 "
            "--nd 16 --f 12 --nm 64 --nmo 96 --nlb 384 --nla 384 --nb-epoch 50 --lr 1e-4 "
            "-o ./out/resisc45_actor_critic").split()
    a = build_parser().parse_args(argv)
    assert (a.agents, a.step, a.cuda, a.run_id) == (16, 16, True, "train_resisc45")
    assert (a.n_b, a.n_a, a.n_m, a.n_m_o, a.n_d, a.n_l_b, a.n_l_a) == (256, 256, 64, 96, 16, 384, 384)
    assert a.learning_rate == 1e-4 and a.f == 12 and a.ft_extr_str == "resisc45"
    assert parse_actions(a.action, a.dim) == [[1, 0], [-1, 0], [0, 1], [0, -1]]
    d = build_parser().parse_args("--run-id x train -o o".split())  # reference defaults
    assert (d.agents, d.step, d.f, d.n_b, d.n_a, d.n_d, d.n_l_b, d.batch_size) == (3, 7, 7, 64, 16, 4, 128, 8)


def test_action_syntax_errors():
    assert parse_actions("[[3, 0], [-3, 0], [0, 3], [0, -3]]", 2)[1] == [-3, 0]
    with pytest.raises(ValueError):
        parse_actions("[1,0],[0,1]", 2)
    with pytest.raises(AssertionError):
        parse_actions("[[1,0,0],[0,1,0]]", 2)


def test_test_and_infer_flags_match_the_reference():
    """reference __main__.py:217-300: the ``test`` and ``infer`` sub-commands and their flags."""
    from marlclassification_amd.__main__ import build_parser

    p = build_parser()
    a = p.parse_args("--run-id r -a 5 --step 9 --cuda test --dataset-path d --json-path j "
                     "--state-dict-path s -o out".split())
    assert (a.main_choice, a.batch_size, a.img_size, a.dataset_path, a.output_dir) == ("test", 8, 28, "d", "out")
    b = p.parse_args("--run-id r infer --images a.png b/*.png --json-path j --state-dict-path s "
                     "--class2idx c.json -o out".split())
    assert b.infer_images == ["a.png", "b/*.png"] and b.class_to_idx == "c.json" and b.output_image_dir == "out"


def test_metric_formatting_and_confusion_png(tmp_path):
    import torch as th

    from marlclassification_amd.metrics import ConfusionMeter, format_metric

    m = ConfusionMeter(3)
    m.add(th.tensor([[0.9, 0.05, 0.05], [0.1, 0.8, 0.1], [0.2, 0.7, 0.1], [0.0, 0.1, 0.9]]), th.tensor([0, 1, 2, 2]))
    assert m.conf_mat().tolist() == [[1, 0, 0], [0, 1, 0], [0, 1, 1]]
    assert format_metric(m.recall(), {"a": 0, "b": 1, "c": 2}) == '"a" : 100.0%, "b" : 100.0%, "c" : 50.0%'
    path = m.save_conf_matrix(3, str(tmp_path), "eval")
    assert path.endswith("confusion_matrix_epoch_3_eval.png") and (tmp_path / "confusion_matrix_epoch_3_eval.png").exists()
