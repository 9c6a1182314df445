"""Pins the oracle against the real reference and writes tests/golden/*.npz.

Runs ONLY in the build container, where the Python reference is importable from
/root/reference (it never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/make_golden.py

For every case it (1) builds the reference's own ModelsWrapper / MultiAgent /
Environment / EpisodeSampler / Trainer objects, (2) runs them on a fixed seed,
(3) runs ``oracle/marl_oracle.py`` on the same weights with the same random
draws injected, (4) REQUIRES bit-identical positions, logits, log-probs,
values, loss scalars, gradients and post-Adam parameters, and only then
(5) writes the inputs and the reference's outputs as a fixture.  A fixture is
data only (inputs + expected outputs); no reference source is stored.

Cases: G1 conftest-size odd dims, G2 MNIST config C1,
G3 shipped MNIST checkpoint replay (5 actions incl. [0,0]), G4 RESISC45 dims
tiny batch, G5 unit known-answer vectors, G6 initial weights under fixed seeds.
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch as th

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from marl_classification.core import Environment, EpisodeSampler, MultiAgent  # noqa: E402
from marl_classification.networks import ModelsWrapper  # noqa: E402
from marl_classification.networks import message as ref_message  # noqa: E402
from marl_classification.networks import vision as ref_vision  # noqa: E402
from marl_classification.training import Trainer  # noqa: E402
from marl_classification.training import functions as ref_fn  # noqa: E402

from oracle import marl_oracle as mo  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF_CNN = {
    "mnist": ref_vision.MnistCnn,
    "resisc45": ref_vision.Resisc45Cnn,
    "aid": ref_vision.AidCnn,
}


def build_reference(cfg: mo.OracleConfig, na: int):
    model = ModelsWrapper(
        REF_CNN[cfg.ft_extr](cfg.window), cfg.n_b, cfg.n_a, cfg.n_m, cfg.n_m_o, cfg.n_d,
        cfg.d, cfg.nb_action, cfg.nb_class, cfg.nlb, cfg.nla,
    )
    return model, MultiAgent(na, model), Environment(cfg.actions, cfg.window)


def uniform_params(cfg: mo.OracleConfig, seed: int) -> mo.Params:
    """Machine-independent init for the big case (th.rand is an exact
    integer->float conversion): U(-b, b), b = sqrt(3 / fan_in) for matrices,
    small non-trivial values for biases and norm affines."""
    g = th.Generator().manual_seed(seed)
    out = {}
    for name, shape in mo.param_shapes(cfg).items():
        r = th.rand(shape, generator=g) * 2 - 1
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[name] = r * (3.0 / fan_in) ** 0.5
        elif name.endswith(".weight"):
            out[name] = 1.0 + 0.1 * r
        else:
            out[name] = 0.1 * r
    return out


def check_equal(name: str, a: th.Tensor, b: th.Tensor) -> None:
    if not th.equal(a, b):
        d = (a.double() - b.double()).abs().max().item()
        raise SystemExit(f"ORACLE != REFERENCE at {name}: max|d| = {d:g}")


def run_case(
    tag: str, cfg: mo.OracleConfig, params: mo.Params, img: th.Tensor, y: th.Tensor,
    na: int, ns: int, seed: int, lr: float, gamma: float, store_params: bool,
    store_grads: bool,
) -> None:
    nb = img.shape[0]
    sizes = list(img.shape[2:])
    model, agents, env = build_reference(cfg, na)
    model.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    assert list(model.state_dict().keys()) == list(params.keys())
    sampler = EpisodeSampler(agents, env, ns)

    # --- reference: rollout only, under no_grad -------------------------------
    th.manual_seed(seed)
    with th.no_grad():
        ref_out = sampler.run_episode(img)

    # --- reference: one full Trainer iteration (rollout, loss, backward, Adam) -
    trainer = Trainer(model, cfg.nb_class, lr, gamma)
    th.manual_seed(seed)
    trainer.train_epoch([(img, y)], 0, sampler)
    ref_grads = {k: v.grad.clone() for k, v in model.named_parameters()}
    ref_after = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # --- oracle with the same draws injected ----------------------------------
    inp = mo.draw_episode_inputs(cfg, na, nb, ns, sizes, seed)
    tr, lo, grads = mo.train_iteration(params, cfg, img, y, inp, ns, gamma)
    check_equal("step_pos", tr.step_pos, ref_out.step_pos)
    check_equal("step_preds", tr.step_preds.detach(), ref_out.step_preds)
    check_equal("step_log_probas", tr.step_log_probas.detach(), ref_out.step_log_probas)
    check_equal("step_values", tr.step_values.detach(), ref_out.step_values)
    for k in params:
        check_equal("grad " + k, grads[k], ref_grads[k])
    after = {k: v.clone() for k, v in params.items()}
    m = {k: th.zeros_like(v) for k, v in params.items()}
    v_ = {k: th.zeros_like(v) for k, v in params.items()}
    mo.adam_step(after, grads, m, v_, 1, lr)
    for k in params:
        check_equal("adam " + k, after[k], ref_after[k])
    # faithful (mask + masked_select) crop gives the same episode
    tr_f = mo.run_episode(params, cfg, img, inp, ns, faithful_crop=True)
    check_equal("faithful step_preds", tr_f.step_preds, ref_out.step_preds)

    # reference loss scalars, recomputed from its own functions on its outputs
    ref_lo = mo.a2c_loss(ref_out.step_preds, ref_out.step_log_probas, ref_out.step_values, y, gamma)
    rw = ref_fn.classification_rewards(ref_out.step_preds, y)
    check_equal("rewards", mo.classification_rewards(ref_out.step_preds, y), rw)
    check_equal("returns", mo.discounted_returns(rw, gamma), ref_fn.discounted_returns(rw, gamma))
    check_equal("loss", lo.loss.detach(), ref_lo.loss)

    fx = {
        "meta_na_nb_ns_seed": np.array([na, nb, ns, seed], dtype=np.int64),
        "meta_lr_gamma": np.array([lr, gamma], dtype=np.float64),
        "img": img.numpy(), "y": y.numpy(),
        "pos0": inp.pos0.numpy(), "h0": inp.h0.numpy(), "c0": inp.c0.numpy(),
        "hc0": inp.hc0.numpy(), "cc0": inp.cc0.numpy(), "q": inp.q.numpy(),
        "ref_step_pos": ref_out.step_pos.numpy(),
        "ref_step_preds": ref_out.step_preds.numpy(),
        "ref_step_log_probas": ref_out.step_log_probas.numpy(),
        "ref_step_values": ref_out.step_values.numpy(),
        "ref_step_actions": tr.step_actions.numpy(),
        "ref_loss": np.array(
            [lo.loss.item(), lo.path.item(), lo.error.item(), lo.critic.item()], dtype=np.float32
        ),
        "params_checksum": np.array(
            [sum(v.double().sum().item() for v in params.values())], dtype=np.float64
        ),
        "grads_abs_sum": np.array(
            [ref_grads[k].double().abs().sum().item() for k in params], dtype=np.float64
        ),
    }
    if store_params:
        for k, v in params.items():
            fx["param/" + k] = v.numpy()
    if store_grads:
        for k in params:
            fx["grad/" + k] = ref_grads[k].numpy()
            fx["after/" + k] = ref_after[k].numpy()
    else:
        # big case: keep a strided sample of every gradient (and its indices)
        for k in params:
            g = ref_grads[k].flatten()
            idx = th.linspace(0, g.numel() - 1, min(g.numel(), 64)).long()
            fx["gradidx/" + k] = idx.numpy()
            fx["gradsample/" + k] = g[idx].numpy()
            fx["aftersample/" + k] = ref_after[k].flatten()[idx].numpy()
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{tag}: oracle == reference (bit-exact); wrote {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB), loss={lo.loss.item():.6f}")


def g1() -> None:
    cfg = mo.OracleConfig("mnist", 12, 23, 22, 21, 20, 19, 10, 24, 25)
    th.manual_seed(1001)
    model, _, _ = build_reference(cfg, 5)  # reference's own init (init.py)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # make norm affines / biases non-trivial so their gradients paths are exercised
    g = th.Generator().manual_seed(7)
    for k, v in params.items():
        if v.dim() == 1:
            v.add_(0.1 * (th.rand(v.shape, generator=g) * 2 - 1))
    img = th.rand(19, 1, 28, 28, generator=th.Generator().manual_seed(0))
    y = th.randint(0, 10, (19,), generator=th.Generator().manual_seed(1))
    run_case("g1_conftest", cfg, params, img, y, 5, 7, 42, 1e-3, 0.99, True, True)


def g2() -> None:
    cfg = mo.OracleConfig("mnist", 6, 64, 64, 16, 24, 8, 10, 96, 96)
    th.manual_seed(1002)
    model, _, _ = build_reference(cfg, 3)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img = th.rand(32, 3, 28, 28, generator=th.Generator().manual_seed(0))
    y = th.randint(0, 10, (32,), generator=th.Generator().manual_seed(1))
    run_case("g2_mnist_c1", cfg, params, img, y, 3, 5, 42, 1e-3, 0.99, True, True)


def g3() -> None:
    cfg = mo.OracleConfig(
        "mnist", 6, 80, 80, 16, 24, 8, 10, 112, 112,
        actions=[[1, 0], [-1, 0], [0, 1], [0, -1], [0, 0]],
    )
    sd = th.load(
        os.path.join(REF, "resources/trained_models/mnist/nn_models_epoch_49.pt"),
        map_location="cpu",
    )
    params = {k: sd[k].clone() for k in mo.param_shapes(cfg)}
    img = th.rand(16, 3, 28, 28, generator=th.Generator().manual_seed(3))
    y = th.randint(0, 10, (16,), generator=th.Generator().manual_seed(4))
    run_case("g3_mnist_ckpt", cfg, params, img, y, 3, 5, 43, 1e-3, 0.99, True, True)


def g4() -> None:
    cfg = mo.OracleConfig("resisc45", 12, 256, 256, 64, 96, 16, 45, 384, 384)
    params = uniform_params(cfg, 2024)
    img = th.rand(2, 3, 256, 256, generator=th.Generator().manual_seed(5))
    y = th.randint(0, 45, (2,), generator=th.Generator().manual_seed(6))
    # image is regenerated from its seed in the test (th.rand is machine independent)
    run_case("g4_resisc_b2", cfg, params, img, y, 16, 16, 44, 1e-4, 0.99, False, False)
    path = os.path.join(OUT, "g4_resisc_b2.npz")
    fx = dict(np.load(path))
    del fx["img"]
    fx["img_seed_shape"] = np.array([5, 2, 3, 256, 256], dtype=np.int64)
    fx["params_uniform_seed"] = np.array([2024], dtype=np.int64)
    np.savez_compressed(path, **fx)
    print(f"g4: image dropped from fixture -> {os.path.getsize(path) / 1024:.0f} KiB")


def g5() -> None:
    """Unit known-answer vectors produced by the reference's own functions."""
    fx = {}
    # crop on a non-square image (private static method reached by its mangled name)
    img = th.rand(3, 2, 17, 23, generator=th.Generator().manual_seed(8))
    pos = th.stack(
        [th.randint(17 - 5, (4, 3), generator=th.Generator().manual_seed(9)),
         th.randint(23 - 5, (4, 3), generator=th.Generator().manual_seed(10))], dim=-1)
    obs = Environment._Environment__observation(img, pos, 5)
    check_equal("crop", mo.crop_patches(img, pos, 5), obs)
    check_equal("crop masked", mo.crop_patches_masked(img, pos, 5), obs)
    fx.update(crop_img=img.numpy(), crop_pos=pos.numpy(), crop_obs=obs.numpy())
    # border transitions, img 10, f 5, from (4, 0)
    table = [[1, 0], [-1, 0], [0, 1], [0, -1], [3, 0], [0, 0]]
    p0 = th.tensor([[[4, 0]]]).repeat(len(table), 1, 1)
    acts = th.arange(len(table)).view(-1, 1)
    newp = Environment._Environment__transition(
        p0.float(), th.tensor(table)[acts], 5, [10, 10]).long()
    check_equal("transition", mo.transition(p0, acts, th.tensor(table), 5, [10, 10]), newp)
    fx.update(tr_table=np.array(table), tr_pos=p0.numpy(), tr_new=newp.numpy())
    # aggregate_messages incl. Na == 1
    m = th.randn(5, 19, 21, generator=th.Generator().manual_seed(11))
    check_equal("agg", mo.aggregate_messages(m), ref_message.aggregate_messages(m))
    check_equal("agg1", mo.aggregate_messages(m[:1]), ref_message.aggregate_messages(m[:1]))
    fx.update(agg_in=m.numpy(), agg_out=ref_message.aggregate_messages(m).numpy())
    # returns / standardize / rewards
    r = th.randn(7, 5, 19, generator=th.Generator().manual_seed(12))
    fx.update(ret_in=r.numpy(), ret_out=ref_fn.discounted_returns(r, 0.99).numpy(),
              std_out=ref_fn.standardize(r).numpy())
    check_equal("returns", mo.discounted_returns(r, 0.99), ref_fn.discounted_returns(r, 0.99))
    check_equal("standardize", mo.standardize(r), ref_fn.standardize(r))
    sp = th.randn(7, 5, 19, 10, generator=th.Generator().manual_seed(13))
    yy = th.randint(0, 10, (19,), generator=th.Generator().manual_seed(14))
    fx.update(rw_preds=sp.numpy(), rw_y=yy.numpy(),
              rw_out=ref_fn.classification_rewards(sp, yy).numpy())
    check_equal("rewards", mo.classification_rewards(sp, yy), ref_fn.classification_rewards(sp, yy))
    # multinomial == argmax(p / q) with q ~ Exp(1) drawn from the same generator
    probs = th.softmax(th.randn(95, 4, generator=th.Generator().manual_seed(15)), -1)
    th.manual_seed(77)
    a_ref = th.multinomial(probs, 1, replacement=True).view(-1)
    th.manual_seed(77)
    q = th.empty(95, 4).exponential_(1)
    check_equal("multinomial", mo.sample_actions(probs, q), a_ref)
    fx.update(mn_probs=probs.numpy(), mn_q=q.numpy(), mn_actions=a_ref.numpy())
    path = os.path.join(OUT, "g5_unit_kats.npz")
    np.savez_compressed(path, **fx)
    print(f"g5: unit KATs == reference; wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def g6() -> None:
    """Initial weights (networks/init.py:6-29 applied by models.py:76): what the
    reference's constructor leaves in every parameter under a fixed torch seed.  The fixture
    holds, per parameter, its float64 sum, float64 sum of |w| and its first eight values - enough
    to pin an init recipe + draw order bit for bit without shipping the tensors."""
    cases = {
        "mnist_conftest": (mo.OracleConfig("mnist", 12, 23, 22, 21, 20, 19, 10, 24, 25), 123),
        "resisc45_readme": (mo.OracleConfig("resisc45", 12, 256, 256, 64, 96, 16, 45, 384, 384), 7),
        "aid_readme": (mo.OracleConfig("aid", 24, 256, 256, 64, 96, 16, 30, 320, 320), 3),
    }
    fx = {}
    for tag, (cfg, seed) in cases.items():
        th.manual_seed(seed)
        model, _, _ = build_reference(cfg, 3)
        sd = model.state_dict()
        assert {k: tuple(v.shape) for k, v in sd.items()} == mo.param_shapes(cfg)
        fx[f"{tag}/seed"] = np.array([seed])
        fx[f"{tag}/names"] = np.array(list(sd))
        fx[f"{tag}/sum"] = np.array([v.double().sum().item() for v in sd.values()])
        fx[f"{tag}/abs"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        fx[f"{tag}/head"] = np.stack([
            np.pad(v.flatten()[:8].numpy(), (0, max(0, 8 - v.numel()))) for v in sd.values()])
    path = os.path.join(OUT, "g6_init.npz")
    np.savez_compressed(path, **fx)
    print(f"g6: reference initial weights under fixed seeds; wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    th.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6"]
    for w in which:
        globals()[w]()
