"""GPU: the reference's Python surface (Environment / MultiAgent / EpisodeSampler /
ModelsWrapper / Trainer) on the HIP path.  The first three tests restate the reference's
own tests (tests/test_environment.py, tests/test_episode.py: shapes and bounds, odd sizes);
the others check numerical parity of the drop-in paths against the golden vectors."""
import math

import pytest
import torch as th
import torch.nn.functional as F

from tests.util import Golden

pytestmark = pytest.mark.gpu

ACTIONS = [[1, 0], [-1, 0], [0, 1], [0, -1]]
NA, NB, NS, F_WIN, NC = 5, 19, 7, 12, 10


def _model(device):
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import MnistCnn

    return ModelsWrapper(MnistCnn(F_WIN), 23, 22, 21, 20, 19, 2, len(ACTIONS), NC, 24, 25).to(device)


def test_environment_reset_step_bounds(device):
    from marlclassification_amd.core import Environment

    env = Environment(ACTIONS, F_WIN)
    x = th.randn(NB, 1, 28, 28, device=device)
    obs = env.reset(x, NA)
    assert env.positions.shape == (NA, NB, 2) and env.positions.dtype == th.int64
    assert obs.shape == (NA, NB, 1, F_WIN, F_WIN)
    # the observation IS the crop at the positions
    p = env.positions.cpu()
    xc = x.cpu()
    for a, b in ((0, 0), (4, 18), (2, 7)):
        assert th.equal(obs[a, b].cpu(), xc[b, :, p[a, b, 0]:p[a, b, 0] + F_WIN, p[a, b, 1]:p[a, b, 1] + F_WIN])
    for _ in range(50):
        obs = env.step(th.randint(env.nb_actions, (NA, NB), device=device))
        assert bool((env.positions >= 0).all())
        assert bool((env.positions + F_WIN <= 28).all())
        assert obs.shape == (NA, NB, 1, F_WIN, F_WIN)
    npos = env.normalized_positions
    assert npos.shape == env.positions.shape and bool((npos >= 0).all()) and bool((npos < 1).all())
    assert th.equal(npos.cpu(), env.positions.cpu().float() / 28.0)


def test_episode_shapes(device):
    from marlclassification_amd.core import Environment, EpisodeSampler, MultiAgent

    model = _model(device)
    sampler = EpisodeSampler(MultiAgent(NA, model), Environment(ACTIONS, F_WIN), NS)
    x = th.randn(NB, 1, 28, 28)  # CPU batch: moved to the model's device like the reference
    with th.no_grad():
        last = sampler.run_episode_get_last_step(x)
    assert last.prediction.shape == (NA, NB, NC) and last.actions_log_probs.shape == (NA, NB)
    out = sampler.run_episode(x)
    assert out.step_preds.shape == (NS, NA, NB, NC)
    assert out.step_log_probas.shape == (NS, NA, NB)
    assert out.step_values.shape == (NS, NA, NB)
    assert out.step_pos.shape == (NS, NA, NB, 2) and out.step_pos.dtype == th.int64
    assert out.step_preds.requires_grad and bool(th.isfinite(out.step_preds).all())
    assert bool((out.step_pos >= 0).all()) and bool((out.step_pos + F_WIN <= 28).all())
    assert bool((out.step_log_probas <= 0).all())


def test_multi_agent_act_step_by_step(device):
    from marlclassification_amd.core import Environment, MultiAgent

    model = _model(device)
    agents, env = MultiAgent(NA, model), Environment(ACTIONS, F_WIN)
    obs = env.reset(th.randn(NB, 1, 28, 28, device=device), NA)
    agents.reset(NB)
    for _ in range(3):
        o = agents.act(obs, env.normalized_positions)
        assert o.actions.shape == (NA, NB) and o.actions.dtype == th.int64
        assert int(o.actions.min()) >= 0 and int(o.actions.max()) < len(ACTIONS)
        assert o.actions_log_probs.shape == (NA, NB) and o.predictions.shape == (NA, NB, NC)
        assert o.values.shape == (NA, NB)
        obs = env.step(o.actions)


def _golden_sampler(g, device):
    from marlclassification_amd.core import Environment, EpisodeSampler, MultiAgent
    from marlclassification_amd.fused import EpisodeDraws
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import MnistCnn

    c = g.cfg
    model = ModelsWrapper(MnistCnn(c.window), c.n_b, c.n_a, c.n_m, c.n_m_o, c.n_d, 2,
                          c.nb_action, c.nb_class, c.nlb, c.nla)
    model.load_state_dict(g.params)
    model.to(device)
    sampler = EpisodeSampler(MultiAgent(g.na, model), Environment(c.actions, c.window), g.ns)
    i = g.inp
    sampler.fixed_draws = EpisodeDraws(*(t.to(device) for t in (i.pos0, i.h0, i.c0, i.hc0, i.cc0, i.q)))
    return model, sampler


def _reference_loss(out, y, gamma):
    """The reference's loss code (training/trainer.py:76-111, functions.py) written with
    torch ops by a *user* of the drop-in autograd path."""
    ns, na, nb, nc = out.step_preds.shape
    predictions = out.step_preds.mean(dim=1).flatten(0, 1)
    error = F.cross_entropy(predictions, y.unsqueeze(0).repeat(ns, 1).flatten(0, 1),
                            reduction="none").unflatten(0, (ns, 1, nb))
    tgt = y[:, None, None].repeat(1, ns, na)
    ce = F.cross_entropy(out.step_preds.permute(2, 3, 0, 1), tgt, reduction="none").permute(1, 2, 0)
    rewards = (math.log(nc) - ce) / math.log(nc)
    t_steps = th.arange(ns, device=y.device).view(ns, 1, 1).float()
    returns = (rewards * gamma**t_steps).flip(dims=(0,)).cumsum(0).flip(dims=(0,)) / gamma**t_steps
    adv = returns - out.step_values
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    path = -out.step_log_probas * adv.detach()
    critic = F.smooth_l1_loss(out.step_values, returns.detach(), reduction="none")
    return th.sum(path + error + critic, 0).mean()


@pytest.mark.parametrize("tag", ["g1_conftest", "g3_mnist_ckpt"])
def test_autograd_drop_in_matches_reference_gradients(device, tag):
    g = Golden(tag)
    model, sampler = _golden_sampler(g, device)
    out = sampler.run_episode(g.img.to(device))
    assert th.equal(out.step_pos.cpu(), g.ref("step_pos"))
    loss = _reference_loss(out, g.y.to(device), g.gamma)
    loss.backward()
    assert abs(loss.item() - g.ref("loss")[0].item()) <= 2e-5 * abs(g.ref("loss")[0].item())
    for k, p in model.named_parameters():
        ref = g.grad(k)
        assert (p.grad.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-7, k


@pytest.mark.parametrize("tag", ["g1_conftest", "g2_mnist_c1"])
def test_trainer_step_matches_reference_update(device, tag):
    from marlclassification_amd.training import Trainer

    g = Golden(tag)
    model, sampler = _golden_sampler(g, device)
    trainer = Trainer(model, g.cfg.nb_class, g.lr, g.gamma)
    trainer.train_epoch([(g.img, g.y)], 0, sampler)
    assert trainer.curr_step == 1
    m = trainer.metrics()
    assert abs(m["loss"] - g.ref("loss")[0].item()) <= 2e-5 * abs(g.ref("loss")[0].item())
    sd = model.state_dict()
    for k in g.params:
        ref_upd = g.after(k) - g.params[k]
        upd = sd[k].cpu() - g.params[k]
        big = g.grad(k).abs() > 1e-6
        if big.any():
            assert (upd[big] - ref_upd[big]).abs().max().item() <= 1e-3 * g.lr, k
    # the updated weights are what the next rollout uses (re-packed)
    with th.no_grad():
        nxt = sampler.run_episode_get_last_step(g.img)
    assert bool(th.isfinite(nxt.prediction).all())


def test_eval_epoch_and_state_dict_roundtrip(device, tmp_path):
    from marlclassification_amd.training import Trainer

    g = Golden("g3_mnist_ckpt")
    model, sampler = _golden_sampler(g, device)
    sampler.fixed_draws = None
    trainer = Trainer(model, g.cfg.nb_class, g.lr, g.gamma)
    cm = trainer.eval_epoch([(g.img, g.y), (g.img, g.y)], 0, sampler)
    assert int(cm.conf_mat().sum()) == 2 * g.nb
    path = tmp_path / "nn_models_epoch_0.pt"
    th.save(model.state_dict(), path)
    sd = th.load(path, map_location="cpu")
    assert list(sd) == list(g.params) and all(th.equal(sd[k], g.params[k]) for k in sd)


def test_uint8_images_on_device_to_tensor(device):
    """a uint8 batch is converted (x / 255, torchvision ToTensor) inside the
    gather kernel; the episode equals the one on the pre-converted fp32 batch bit for bit."""
    g = Golden("g1_conftest")
    model, sampler = _golden_sampler(g, device)
    img_u8 = (g.img * 255).round().to(th.uint8)
    img_f = img_u8.to(th.float32).div(255)
    with th.no_grad():
        a = sampler.run_episode(img_u8.to(device))
        b = sampler.run_episode(img_f.to(device))
    assert th.equal(a.step_pos, b.step_pos)
    assert th.equal(a.step_preds, b.step_preds) and th.equal(a.step_values, b.step_values)


def test_train_main_cli_end_to_end(device, tmp_path):
    """the reference's command line drives the HIP path; the output tree holds
    marl.json, class_to_idx.json and per-epoch state dicts with reference keys."""
    import json

    from marlclassification_amd.__main__ import main
    from oracle import marl_oracle as mo

    out = tmp_path / "run"
    argv = (f"-a 3 --step 5 --cuda --run-id smoke train --ft-extr mnist --f 6 --img-size 28 --nb 64 "
            f"--na 64 --nm 16 --nmo 24 --nd 8 --nlb 96 --nla 96 --batch-size 16 --nb-epoch 2 "
            f"--lr 1e-3 --res-folder synthetic -o {out}").split()
    main(argv)
    cfg = json.loads((out / "marl.json").read_text())
    assert cfg["window_size"] == 6 and cfg["actions"] == [[1, 0], [-1, 0], [0, 1], [0, -1]]
    assert (out / "class_to_idx.json").exists()
    sd = th.load(out / "models" / "nn_models_epoch_1.pt", map_location="cpu")
    ocfg = mo.OracleConfig("mnist", 6, 64, 64, 16, 24, 8, 10, 96, 96)
    assert {k: tuple(v.shape) for k, v in sd.items()} == mo.param_shapes(ocfg)
    assert all(bool(th.isfinite(v).all()) for v in sd.values())
    sd0 = th.load(out / "models" / "nn_models_epoch_0.pt", map_location="cpu")
    assert any(not th.equal(sd[k], sd0[k]) for k in sd), "weights did not change between epochs"


def test_cli_train_test_infer_on_an_image_folder(device, tmp_path, monkeypatch, capsys):
    """the reference's three modes end to end on a (tiny) image-folder dataset
    laid out like the reference's resources directory: ``train`` (dataset resolved under
    ``downloaded/mnist_png/all_png``, per-epoch state dicts, confusion-matrix PNG, step GIF),
    ``test`` (confusion matrix, precision / recall) and ``infer`` (frames + GIF per image)."""
    import json

    import numpy as np
    from PIL import Image

    from marlclassification_amd.__main__ import main

    res = tmp_path / "resources"
    root = res / "downloaded" / "mnist_png" / "all_png"
    rng = np.random.default_rng(0)
    for c in range(3):
        (root / f"class{c}").mkdir(parents=True)
        for k in range(12):
            arr = rng.integers(0, 256, (28, 28), dtype=np.uint8)
            arr[4 * c: 4 * c + 8] = 255  # a class-dependent bright band
            Image.fromarray(arr).save(root / f"class{c}" / f"img{k}.png")
    out = tmp_path / "run"
    common = "-a 3 --step 4 --cuda --run-id cli"
    main((f"{common} train --ft-extr mnist --f 6 --img-size 28 --nb-class 3 --nb 32 --na 32 --nm 8 "
          f"--nmo 12 --nd 8 --nlb 48 --nla 48 --batch-size 8 --nb-epoch 1 --lr 1e-3 --res-folder {res} "
          f"-o {out}").split())
    assert (out / "models" / "nn_models_epoch_0.pt").exists()
    assert (out / "confusion_matrix_epoch_0_eval.png").exists()
    assert (out / "animated_gif.gif").exists() and (out / "pred_step_3.png").exists()
    assert json.loads((out / "class_to_idx.json").read_text()) == {"class0": 0, "class1": 1, "class2": 2}
    # the same command streamed through DataLoader workers instead of the HBM-resident image set
    monkeypatch.setenv("MARL_RESIDENT_GB", "0")
    monkeypatch.setenv("MARL_LOADER_WORKERS", "2")
    out_s = tmp_path / "run_streamed"
    main((f"{common} train --ft-extr mnist --f 6 --img-size 28 --nb-class 3 --nb 32 --na 32 --nm 8 "
          f"--nmo 12 --nd 8 --nlb 48 --nla 48 --batch-size 8 --nb-epoch 1 --lr 1e-3 --res-folder {res} "
          f"-o {out_s}").split())
    assert (out_s / "models" / "nn_models_epoch_0.pt").exists()
    assert "streamed" in capsys.readouterr().out
    monkeypatch.delenv("MARL_RESIDENT_GB")
    # a wrong resources folder is an error, not a silent synthetic run
    with pytest.raises(NotADirectoryError):
        main((f"{common} train --ft-extr mnist --f 6 --img-size 28 --nb-class 3 --res-folder {tmp_path / 'nope'} "
              f"-o {out}").split())
    test_out = tmp_path / "test_out"
    main((f"{common} test --batch-size 8 --dataset-path {root} --img-size 28 --json-path {out / 'marl.json'} "
          f"--state-dict-path {out / 'models' / 'nn_models_epoch_0.pt'} -o {test_out}").split())
    assert (test_out / "confusion_matrix_epoch_0_test.png").exists()
    inf_out = tmp_path / "infer_out"
    main((f"{common} infer --images {root / 'class1' / 'img0.png'} {root / 'class2' / 'img*.png'} "
          f"--json-path {out / 'marl.json'} --state-dict-path {out / 'models' / 'nn_models_epoch_0.pt'} "
          f"--class2idx {out / 'class_to_idx.json'} -o {inf_out}").split())
    assert (inf_out / "img0.png" / "animated_gif.gif").exists()
    assert (inf_out / "img3.png" / "pred_step_0.png").exists() and (inf_out / "img3.png" / "info.txt").exists()


def test_two_live_episodes_accumulate_like_autograd(device):
    """The reference's autograd lets ``(loss1 + loss2).backward()`` run over two ``run_episode`` calls of one
    model (core/episode.py:84 there keeps every rollout's graph alive).  Every episode here owns its
    saved activations, so: two rollouts (different draws, different batch sizes), ONE backward of the
    summed loss == the sum of the two separate gradients, in either order; a backward after the
    weights moved is refused."""
    g = Golden("g1_conftest")
    model, sampler = _golden_sampler(g, device)
    from marlclassification_amd.fused import EpisodeDraws

    img, y = g.img.to(device), g.y.to(device)
    i = g.inp
    d_all = sampler.fixed_draws
    d_7 = EpisodeDraws(*(t.to(device).contiguous() for t in (i.pos0[:, 5:12], i.h0[:, 5:12], i.c0[:, 5:12],
                                                            i.hc0[:, 5:12], i.cc0[:, 5:12], i.q[:, :, 5:12])))

    def episode(small):
        sampler.fixed_draws = d_7 if small else d_all
        return sampler.run_episode(img[5:12] if small else img)

    def grads_of(fn):
        model.zero_grad(set_to_none=True)
        fn()
        return {k: p.grad.clone() for k, p in model.named_parameters()}

    def separate():
        _reference_loss(episode(False), y, g.gamma).backward()
        _reference_loss(episode(True), y[5:12], g.gamma).backward()

    def summed():
        o1 = episode(False)
        o2 = episode(True)  # a second live episode, another batch size
        (_reference_loss(o1, y, g.gamma) + _reference_loss(o2, y[5:12], g.gamma)).backward()

    def reversed_order():
        o1 = episode(False)
        o2 = episode(True)
        _reference_loss(o2, y[5:12], g.gamma).backward()
        _reference_loss(o1, y, g.gamma).backward()

    a, b, c = grads_of(separate), grads_of(summed), grads_of(reversed_order)
    for k in a:
        scale = a[k].abs().max().item() + 1e-12
        assert (a[k] - b[k]).abs().max().item() <= 2e-6 * scale, k
        assert (a[k] - c[k]).abs().max().item() <= 2e-6 * scale, k
    # stale weights: refuse instead of differentiating through other weights than the rollout used
    o1 = episode(False)
    with th.no_grad():
        next(model.parameters()).add_(1e-3)
    episode(True)  # (re-packs the modified weights)
    with pytest.raises(RuntimeError, match="weights were modified"):
        _reference_loss(o1, y, g.gamma).backward()
