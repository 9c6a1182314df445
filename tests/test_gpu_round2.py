"""GPU: parity at the benched size and the edges the first round left open,
the library's own random draws, hipGraph replay and the data-parallel trainer wiring.

Everything goes through the C ABI.  Tolerances as in test_gpu_episode.py: integer tensors
bit-exact, fp32 outputs 1e-5, gradients 1e-4 of the tensor's scale."""
import json
import math
import os
import socket

import pytest
import torch as th

from oracle import marl_oracle as mo
from tests.util import Golden, model_spec, uniform_params

pytestmark = pytest.mark.gpu

ATOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _maxerr(a, b):
    return (a.detach().cpu().double() - b.double()).abs().max().item()


def _engine(cfg, device, na, nb, ns, shape, params, img_u8=False):
    from marlclassification_amd.engine import HipEngine

    eng = HipEngine(model_spec(cfg), device)
    eng.configure(na, nb, ns, shape, img_u8=img_u8)
    eng.pack({k: v.to(device) for k, v in params.items()})
    return eng


def _oracle_case(cfg, na, nb, ns, shape, seed=7):
    params = uniform_params(cfg, seed)
    img = th.rand(nb, *shape, generator=th.Generator().manual_seed(seed + 4))
    y = th.randint(0, cfg.nb_class, (nb,), generator=th.Generator().manual_seed(seed + 5))
    inp = mo.draw_episode_inputs(cfg, na, nb, ns, shape[1:], seed + 6)
    return params, img, y, inp


def _check_against_oracle(eng, cfg, device, params, img, y, inp, ns, img_dev=None, free_running=True):
    """Teacher-forced episode + loss + every gradient against the oracle; returns achieved errors."""
    tr, lo, grads = mo.train_iteration(params, cfg, img, y, inp, ns, 0.99)
    args = [t.to(device) for t in (inp.pos0, inp.h0, inp.c0, inp.hc0, inp.cc0, inp.q)]
    img_dev = img.to(device) if img_dev is None else img_dev
    out = eng.episode_forward(img_dev, *args, tr.step_actions.to(device), True)
    assert th.equal(out.step_pos.cpu(), tr.step_pos), "agent positions differ"
    errs = {"preds": _maxerr(out.step_preds, tr.step_preds), "logp": _maxerr(out.step_log_probas, tr.step_log_probas),
            "values": _maxerr(out.step_values, tr.step_values)}
    assert max(errs.values()) <= ATOL, errs
    gp, gl, gv, sc, _ = eng.a2c_loss(out, y.to(device), 0.99)
    assert abs(sc[0].item() - lo.loss.item()) <= 5e-5 * max(1.0, abs(lo.loss.item()))
    g_out = {k: th.full_like(v, float("nan"), device=device) for k, v in params.items()}
    eng.episode_backward(gp, gl, gv, g_out)
    bad, worst = {}, 0.0
    for k, ref in grads.items():
        err = _maxerr(g_out[k], ref)
        rel = err / (ref.abs().max().item() + 1e-30)
        worst = max(worst, rel if ref.abs().max().item() > 1e-12 else 0.0)
        if not err <= 1e-4 * ref.abs().max().item() + 1e-7:
            bad[k.replace("_ModelsWrapper__", "")] = "%.2e/%.2e" % (err, ref.abs().max().item())
    assert not bad, "\n".join(f"{k}: {v}" for k, v in bad.items())
    if free_running:
        free = eng.episode_forward(img_dev, *args, None, False)
        nflip = (free.step_actions.cpu() != tr.step_actions).sum().item()
        allowed = flip_budget(tr, inp, errs["logp"])
        FLIPS.append({"numel": tr.step_actions.numel(), "flips": nflip, "allowed": allowed})
        assert nflip <= allowed, f"{nflip} sampled actions differ (budget {allowed})"
    errs["grad_rel"] = worst
    return errs


FLIPS = []  # (numel, flips, allowed) of every oracle-only free-running comparison of this session


def flip_budget(tr, inp, logp_err):
    """How many sampled indices MAY differ from the oracle's: a sample is argmax_k p_k / q_k
    (core/agent.py:53-55); it can only flip where the two largest ratios are closer than the
    relative error of the probabilities.  Counted on the oracle's own probabilities and noise
    with 4x the measured log-prob error (>= 2e-6) as that relative error: 0 for almost every case."""
    eps = 4.0 * max(logp_err, 5e-7)
    if not hasattr(tr, "step_probs") or tr.step_probs is None:
        return 0
    ratio = tr.step_probs.double() / inp.q.double().view_as(tr.step_probs)
    top2 = ratio.topk(2, dim=-1).values
    return int(((top2[..., 0] - top2[..., 1]) <= eps * top2[..., 0]).sum().item())


# ---- (a) the benched size: B = 256, R = 4096 rows, the 128x128 / grouped / split-K plans -----
def test_benched_size_replicas_match_resisc_goldens(device):
    """The G4 fixture (RESISC45 dims, 2 images, generated from the real reference) tiled 128x
    along the batch: every replica must reproduce the reference's positions and sampled actions
    bit for bit and its logits / values within 1e-5, at the shapes bench.py runs (R = 4096 rows,
    NR = 65536).  The loss is finished with the advantage statistics of the un-tiled batch
    (phase 2 of marl_a2c_loss_fwd_bwd), so the tiled batch's gradient equals the fixture's:
    the sampled reference gradients are checked at the benched size too."""
    g = Golden("g4_resisc_b2")
    rep = 128
    eng1 = _engine(g.cfg, device, g.na, g.nb, g.ns, g.img.shape[1:], g.params)
    i = g.inp
    small = [t.to(device) for t in (i.pos0, i.h0, i.c0, i.hc0, i.cc0, i.q)]
    out1 = eng1.episode_forward(g.img.to(device), *small, None, True)
    _, _, _, _, stats = eng1.a2c_loss(out1, g.y.to(device), g.gamma, phase=1)
    stats = stats.clone()
    del eng1

    def tile(t, dim):
        return t.repeat_interleave(rep, dim=dim) if False else th.cat([t] * rep, dim=dim)

    nb = g.nb * rep
    eng = _engine(g.cfg, device, g.na, nb, g.ns, g.img.shape[1:], g.params)
    big = [tile(i.pos0, 1), tile(i.h0, 1), tile(i.c0, 1), tile(i.hc0, 1), tile(i.cc0, 1), tile(i.q, 2)]
    img = tile(g.img, 0).to(device)
    y = tile(g.y, 0).to(device)
    out = eng.episode_forward(img, *[t.to(device) for t in big], None, True)

    def replicas(t, bdim):  # [.., nb, ..] -> [rep, .., g.nb, ..]
        shape = list(t.shape)
        shape[bdim:bdim + 1] = [rep, g.nb]
        return t.reshape(shape).movedim(bdim, 0)

    pos = replicas(out.step_pos.cpu(), 2)
    act = replicas(out.step_actions.cpu(), 2)
    assert th.equal(pos, g.ref("step_pos").expand_as(pos)), "positions of a replica differ"
    assert th.equal(act, g.ref("step_actions").expand_as(act)), "sampled actions of a replica differ"
    for name, got, ref in (("preds", out.step_preds, g.ref("step_preds")),
                           ("logp", out.step_log_probas, g.ref("step_log_probas")),
                           ("values", out.step_values, g.ref("step_values"))):
        r = replicas(got.cpu(), 2)
        assert (r - ref).abs().max().item() <= ATOL, name
        assert th.equal(r[0], r[rep - 1]), f"{name}: replicas are not bit-identical"
    # loss with the un-tiled batch's statistics: stats = (n, sum, sum of squares) scale with rep
    stats_big = stats * rep
    bufs = None
    gp, gl, gv, sc, _ = eng.a2c_loss(out, y, g.gamma, phase=1)
    st_now = _.clone()
    assert th.allclose(st_now, stats_big, rtol=1e-9), (st_now, stats_big)
    # standardize over the tiled batch uses n*rep - 1 in the unbiased variance; inject the
    # fixture's own mean / std by passing ITS statistics: n, sum, sumsq of the 2-image batch
    _.copy_(stats)
    gp, gl, gv, sc, _ = eng.a2c_loss(out, y, g.gamma, phase=2, bufs=(gp, gl, gv, sc, _))
    assert th.allclose(sc.cpu(), g.ref("loss"), rtol=5e-5, atol=5e-5), (sc.cpu(), g.ref("loss"))
    grads = {k: th.zeros_like(v, device=device) for k, v in g.params.items()}
    eng.episode_backward(gp, gl, gv, grads)
    bad = {}
    for n_, k in enumerate(g.params):
        idx = th.from_numpy(g.z["gradidx/" + k])
        ref = th.from_numpy(g.z["gradsample/" + k])
        got = grads[k].flatten()[idx.to(device)].cpu()
        scale = float(g.z["grads_abs_sum"][n_]) / max(1, g.params[k].numel())
        err = (got - ref).abs().max().item()
        if not err <= 1e-4 * max(ref.abs().max().item(), scale) + 1e-7:
            bad[k] = (err, ref.abs().max().item())
    assert not bad, bad


# ---- (b) Na == 1: aggregate_messages returns zeros (networks/message.py:14-15) ----------------
def test_single_agent_episode_matches_oracle(device):
    cfg = mo.OracleConfig("mnist", 6, 32, 32, 8, 12, 8, 10, 48, 48)
    na, nb, ns, shape = 1, 24, 4, (3, 28, 28)
    params, img, y, inp = _oracle_case(cfg, na, nb, ns, shape)
    eng = _engine(cfg, device, na, nb, ns, shape, params)
    _check_against_oracle(eng, cfg, device, params, img, y, inp, ns)


# ---- (e) WorldStrat: 5 conv layers, 256 channels (networks/vision.py:80-86) -------------------
def test_worldstrat_five_layer_cnn_matches_oracle(device):
    cfg = mo.OracleConfig("worldstrat", 32, 48, 40, 16, 24, 8, 6, 56, 64,
                          actions=[[2, 0], [-2, 0], [0, 2], [0, -2]])
    na, nb, ns, shape = 3, 2, 3, (3, 72, 80)
    params, img, y, inp = _oracle_case(cfg, na, nb, ns, shape, seed=21)
    eng = _engine(cfg, device, na, nb, ns, shape, params)
    _check_against_oracle(eng, cfg, device, params, img, y, inp, ns)


# ---- the published checkpoints' widths ("nothing may be hard-coded to the README dims") --
@pytest.mark.parametrize("dims", [(512, 512, 384, 384), (768, 512, 768, 758)])
def test_published_checkpoint_widths_match_oracle(device, dims):
    """resources/trained_models use n_b = 512 / 768 and head widths up to 1024: the wide belief
    state runs through the chained panel kernels (LDS tail of the backward chain, or its fallback
    when the extra panel does not fit), the 758-wide policy layer through the unfused path."""
    n_b, n_a, nlb, nla = dims
    cfg = mo.OracleConfig("resisc45", 12, n_b, n_a, 64, 96, 16, 45, nlb, nla)
    na, nb, ns, shape = 4, 3, 3, (3, 40, 48)
    params, img, y, inp = _oracle_case(cfg, na, nb, ns, shape, seed=33)
    eng = _engine(cfg, device, na, nb, ns, shape, params)
    _check_against_oracle(eng, cfg, device, params, img, y, inp, ns)


# ---- (f) uint8 images: ToTensor inside the kernels vs the oracle on x / 255 -------------------
def test_uint8_episode_matches_oracle_on_scaled_images(device):
    g = Golden("g1_conftest")
    img_u8 = (g.img * 255).round().to(th.uint8)
    img_f = img_u8.to(th.float32) / 255  # what torchvision's ToTensor yields
    eng = _engine(g.cfg, device, g.na, g.nb, g.ns, g.img.shape[1:], g.params, img_u8=True)
    _check_against_oracle(eng, g.cfg, device, g.params, img_f, g.y, g.inp, g.ns, img_dev=img_u8.to(device))


# ---- (c) loss phases 1 -> exchange -> 2 == phase 0 on the whole batch --------------------------
def test_loss_phases_on_two_half_batches_equal_the_big_batch(device):
    g = Golden("g2_mnist_c1")
    h = g.nb // 2
    ref = {k: g.ref(k) for k in ("step_preds", "step_log_probas", "step_values")}
    from marlclassification_amd.engine import EpisodeTensors

    def outs(lo, hi):
        return EpisodeTensors(*(ref[k][:, :, lo:hi].contiguous().to(device) for k in
                                ("step_preds", "step_log_probas", "step_values")), None, None)

    eng_big = _engine(g.cfg, device, g.na, g.nb, g.ns, g.img.shape[1:], g.params)
    gp0, gl0, gv0, sc0, st0 = eng_big.a2c_loss(outs(0, g.nb), g.y.to(device), g.gamma, 0)
    # one engine per shard, like one process per GPU: phase 2 continues from the advantages phase 1
    # left in THAT engine's workspace
    engs = [_engine(g.cfg, device, g.na, h, g.ns, g.img.shape[1:], g.params) for _ in range(2)]
    halves, stats = [], []
    for eng_half, lo in zip(engs, (0, h)):
        o = outs(lo, lo + h)
        bufs = eng_half.a2c_loss(o, g.y[lo:lo + h].to(device), g.gamma, 1)
        stats.append(bufs[4].clone())
        halves.append((o, bufs))
    tot = stats[0] + stats[1]  # the all-reduce of the exact-standardize exchange
    assert th.allclose(tot.cpu(), st0.cpu(), rtol=1e-12), (tot, st0)
    sc_sum = th.zeros(4)
    for k, lo in enumerate((0, h)):
        o, bufs = halves[k]
        bufs[4].copy_(tot)
        gp, gl, gv, sc, _ = engs[k].a2c_loss(o, g.y[lo:lo + h].to(device), g.gamma, 2, bufs)
        # the per-shard mean over (Na, Nb/2): gradients w.r.t. outputs are 2x the big batch's
        for a, b, n in ((gp, gp0, "g_preds"), (gl, gl0, "g_logp"), (gv, gv0, "g_values")):
            bb = b[:, :, lo:lo + h] * 2
            assert _maxerr(a, bb.cpu()) <= 2e-6 * bb.abs().max().item() + 1e-9, n
        sc_sum += sc.cpu()
    assert th.allclose(sc_sum / 2, sc0.cpu(), rtol=2e-6, atol=2e-6)


# ---- (d) MultiAgent.act: sampled actions / log-probs with injected noise ----------------------
def test_multi_agent_act_sampling_matches_oracle(device):
    from marlclassification_amd.core import Environment, MultiAgent
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.models import RecurrentOutput
    from marlclassification_amd.networks.vision import MnistCnn

    g = Golden("g1_conftest")
    c = g.cfg
    model = ModelsWrapper(MnistCnn(c.window), c.n_b, c.n_a, c.n_m, c.n_m_o, c.n_d, 2, c.nb_action,
                          c.nb_class, c.nlb, c.nla)
    model.load_state_dict(g.params)
    model.to(device)
    agents, env = MultiAgent(g.na, model), Environment(c.actions, c.window)
    i = g.inp
    obs = env.place(g.img.to(device), g.na, positions=i.pos0.to(device))
    obs = env.observe()
    agents.reset(g.nb)
    # inject the fixture's initial state and noise
    agents._MultiAgent__hidden = RecurrentOutput(*(t.to(device) for t in (i.h0, i.c0, i.hc0, i.cc0)))
    agents.fixed_noise = i.q[0].to(device)
    o = agents.act(obs, env.normalized_positions)
    sizes = list(g.img.shape[2:])
    so = mo.step_forward(g.params, c, mo.crop_patches(g.img, i.pos0, c.window),
                         th.zeros(g.na, g.nb, c.n_m), mo.normalized_positions(i.pos0, sizes), i.h0, i.c0,
                         i.hc0, i.cc0)
    a_ref = mo.sample_actions(so.probs.flatten(0, 1), i.q[0].flatten(0, 1)).view(g.na, g.nb)
    assert th.equal(o.actions.cpu(), a_ref)
    assert th.equal(o.actions.cpu(), g.ref("step_actions")[0])
    lp_ref = th.log(so.probs.gather(-1, a_ref.unsqueeze(-1)).squeeze(-1))
    assert _maxerr(o.actions_log_probs, lp_ref) <= ATOL
    assert _maxerr(o.predictions, so.preds) <= ATOL and _maxerr(o.values, so.values) <= ATOL


# ---- the library's own draws ---------------------------------------------------------------------
def test_library_draws_have_the_reference_distributions(device):
    cfg = mo.OracleConfig("mnist", 6, 64, 48, 16, 24, 8, 10, 96, 96)
    eng = _engine(cfg, device, 8, 512, 5, (3, 28, 40), uniform_params(cfg, 1))
    pos0, h0, c0, hc0, cc0, noise = eng.draw_episode(1234, 0, with_noise=True)
    again = eng.draw_episode(1234, 0, with_noise=True)
    other = eng.draw_episode(1234, 1, with_noise=True)
    for a, b, c in zip((pos0, h0, c0, hc0, cc0, noise), again, other):
        assert th.equal(a, b), "same (seed, offset) must give the same draws"
        assert not th.equal(a, c), "another offset must give other draws"
    # positions: uniform integers in [0, size - f) per dimension (environment.py:33-43)
    for d, size in enumerate((28, 40)):
        p = pos0[..., d].flatten().cpu()
        span = size - 6
        assert int(p.min()) >= 0 and int(p.max()) == span - 1
        cnt = th.bincount(p, minlength=span).double()
        chi2 = ((cnt - cnt.mean()) ** 2 / cnt.mean()).sum().item()
        assert chi2 < span + 6 * math.sqrt(2 * span), (d, chi2)  # ~ mean + 6 sigma of chi2(span-1)
    # states: standard normal (models.py:148-159), independent streams
    for t in (h0, c0, hc0, cc0):
        x = t.double().flatten().cpu()
        n = x.numel()
        assert abs(x.mean().item()) < 6 / math.sqrt(n)
        assert abs(x.var().item() - 1) < 6 * math.sqrt(2 / n)
        assert abs((x ** 4).mean().item() - 3) < 6 * math.sqrt(96 / n)
    assert abs(th.corrcoef(th.stack([h0.flatten(), c0.flatten()]))[0, 1].item()) < 6 / math.sqrt(h0.numel())
    # Exp(1) noise: mean 1, variance 1, strictly positive
    q = noise.double().flatten().cpu()
    assert float(q.min()) > 0 and abs(q.mean().item() - 1) < 6 / math.sqrt(q.numel())
    assert abs(q.var().item() - 1) < 6 * math.sqrt(8 / q.numel())


def test_in_kernel_sampling_follows_the_policy_probabilities(device):
    """Perf-mode sampling (noise == NULL): action frequencies over many rows that share one
    probability vector must match it (chi-square), and equal (seed, offset) must replay."""
    g = Golden("g1_conftest")
    c = g.cfg
    na, nb = 8, 2048
    eng = _engine(c, device, na, nb, 1, (1, c.window + 1, c.window + 1), g.params)
    gen = th.Generator().manual_seed(3)
    one = lambda *s: th.randn(*s, generator=gen)  # noqa: E731
    obs = one(1, 1, 1, c.window, c.window).expand(na, nb, -1, -1, -1).contiguous()
    ins = [obs, th.zeros(na, nb, c.n_m), one(1, 1, 2).abs().fmod(1).expand(na, nb, -1).contiguous(),
           one(1, 1, c.n_b).expand(na, nb, -1).contiguous(), one(1, 1, c.n_b).expand(na, nb, -1).contiguous(),
           one(1, 1, c.n_a).expand(na, nb, -1).contiguous(), one(1, 1, c.n_a).expand(na, nb, -1).contiguous()]
    ins = [t.to(device) for t in ins]
    res = eng.step_forward(*ins, rng=(99, 5))
    probs, actions, logp = res[0], res[8], res[9]
    p = probs[0, 0].double().cpu()
    assert (probs.view(-1, p.numel()) - probs[0, 0]).abs().max().item() <= 1e-6  # identical rows
    n = actions.numel()
    cnt = th.bincount(actions.flatten().cpu(), minlength=p.numel()).double()
    chi2 = ((cnt - n * p) ** 2 / (n * p)).sum().item()
    k = p.numel() - 1
    assert chi2 < k + 6 * math.sqrt(2 * k) + 6, (chi2, cnt / n, p)
    assert _maxerr(logp, th.log(probs.gather(-1, actions.unsqueeze(-1)).squeeze(-1)).cpu()) <= ATOL
    res2 = eng.step_forward(*ins, rng=(99, 5))
    res3 = eng.step_forward(*ins, rng=(99, 6))
    assert th.equal(res2[8], actions) and not th.equal(res3[8], actions)
    # whole episodes in perf mode: reproducible for a (seed, offset) pair, positions in bounds
    eng2 = _engine(g.cfg, device, g.na, g.nb, g.ns, g.img.shape[1:], g.params)
    d = eng2.draw_episode(5, 0)
    outs = [eng2.episode_forward(g.img.to(device), *d[:5], None, None, False, rng=(5, off)) for off in (0, 0, 1)]
    assert th.equal(outs[0].step_actions, outs[1].step_actions)
    assert th.equal(outs[0].step_preds, outs[1].step_preds)
    assert not th.equal(outs[0].step_actions, outs[2].step_actions)
    assert bool((outs[0].step_pos >= 0).all()) and bool((outs[0].step_pos + c.window <= 28).all())


# ---- hipGraph replay of the whole iteration ------------------------------------------------------
@pytest.mark.parametrize("tag", ["g2_mnist_c1"])
def test_graph_replay_equals_eager_iterations(device, tag):
    from marlclassification_amd.fused import FlatParams, FusedA2C, draw_episode_device

    g = Golden(tag)
    img, y = g.img.to(device), g.y.to(device)
    finals = []
    for use_graph in (False, True):
        eng = _engine(g.cfg, device, g.na, g.nb, g.ns, g.img.shape[1:], g.params)
        flat = FlatParams(mo.param_shapes(g.cfg), device)
        flat.load(g.params)
        fa = FusedA2C(eng, flat, 1e-3, g.gamma, use_graph=use_graph)
        losses = []
        for it in range(5):
            if use_graph:
                out, sc = fa.iteration_graph(img, y, 77, it)
            else:
                out, sc = fa.iteration(img, y, draw_episode_device(eng, 77, it))
            losses.append(sc.clone())
        th.cuda.synchronize()
        finals.append((flat.params.clone(), th.stack(losses), out.step_pos.clone(), flat.step))
    (p0, l0, pos0, s0), (p1, l1, pos1, s1) = finals
    assert s0 == s1 == 5
    assert th.equal(pos0, pos1), "replayed episodes must draw the same positions / actions"
    assert th.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0, l1)
    # the only difference allowed: Adam's bias correction computed on the device (1 ulp)
    assert (p0 - p1).abs().max().item() <= 1e-6 * p0.abs().max().item()


# ---- a stale backward must fail loudly ------------------------------------------------
def test_backward_of_an_overwritten_episode_raises(device):
    """The engine-level calls share ONE training workspace (the fused Trainer's path): a backward for a
    rollout that a later rollout overwrote is refused.  (Episodes run through autograd own their
    workspace since round 3 - tests/test_gpu_api.py::test_two_live_episodes_accumulate_like_autograd.)"""
    from tests.test_gpu_api import _golden_sampler

    g = Golden("g1_conftest")
    model, sampler = _golden_sampler(g, device)
    img = g.img.to(device)
    eng, out1 = sampler.run_episode_raw(img, True)
    gen1 = eng.fwd_generation
    gp, gl, gv, _, _ = eng.a2c_loss(out1, g.y.to(device), g.gamma)
    eng, out2 = sampler.run_episode_raw(img, True)  # overwrites the engine's training workspace
    grads = {k: th.empty_like(p) for k, p in model.named_parameters()}
    with pytest.raises(RuntimeError, match="overwritten"):
        eng.episode_backward(gp, gl, gv, grads, generation=gen1)
    gp, gl, gv, _, _ = eng.a2c_loss(out2, g.y.to(device), g.gamma)
    eng.episode_backward(gp, gl, gv, grads, generation=eng.fwd_generation)  # the live episode still works
    assert all(bool(th.isfinite(v).all()) for v in grads.values())


# ---- data-parallel trainer: HIP path + all-reduce + exact standardize, 2 ranks over gloo --------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, exact, out_q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marlclassification_amd.fused import EpisodeDraws
    from marlclassification_amd.parallel import GradAllReduce, broadcast_parameters, shard_bounds
    from marlclassification_amd.training import Trainer
    from tests.test_gpu_api import _golden_sampler

    device = th.device("cuda:0")  # both ranks share the one GPU of the test box
    g = Golden("g2_mnist_c1")
    model, sampler = _golden_sampler(g, device)
    broadcast_parameters(model.flat_state().params)
    lo, hi = shard_bounds(g.nb, rank, world)
    i = g.inp
    sampler.fixed_draws = EpisodeDraws(*(t.to(device) for t in (
        i.pos0[:, lo:hi].contiguous(), i.h0[:, lo:hi].contiguous(), i.c0[:, lo:hi].contiguous(),
        i.hc0[:, lo:hi].contiguous(), i.cc0[:, lo:hi].contiguous(), i.q[:, :, lo:hi].contiguous())))
    trainer = Trainer(model, g.cfg.nb_class, g.lr, g.gamma, allreduce=GradAllReduce(world),
                      exact_standardize_group=dist.group.WORLD if exact else None)
    trainer.train_step(g.img[lo:hi], g.y[lo:hi], sampler)
    th.cuda.synchronize()
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}  # numpy: plain pickles
    grads = model.flat_state().grads.cpu().numpy()
    if rank == 0:
        out_q.put((sd, grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exact", [True, False])
def test_two_rank_hip_trainer_matches_big_batch_update(device, exact):
    """Two processes, each running the HIP Trainer on half of the G2 batch with the flat-gradient
    all-reduce (+ the exact-standardize exchange): with global statistics the averaged gradient
    and the Adam update equal the REFERENCE's big-batch gradient / update of the fixture; with
    per-shard statistics they equal the mean of the oracle's shard gradients."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, exact, q)) for r in range(2)]
    for p in procs:
        p.start()
    sd, flat_grads = q.get(timeout=600)
    sd = {k: th.from_numpy(v) for k, v in sd.items()}
    flat_grads = th.from_numpy(flat_grads)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = Golden("g2_mnist_c1")
    names = list(g.params)
    if exact:
        ref_g = {k: g.grad(k) for k in names}
    else:
        parts = []
        i = g.inp
        for lo, hi in ((0, g.nb // 2), (g.nb // 2, g.nb)):
            inp = mo.EpisodeInputs(i.pos0[:, lo:hi], i.h0[:, lo:hi], i.c0[:, lo:hi], i.hc0[:, lo:hi],
                                   i.cc0[:, lo:hi], i.q[:, :, lo:hi])
            _, _, gr = mo.train_iteration(g.params, g.cfg, g.img[lo:hi], g.y[lo:hi], inp, g.ns, g.gamma)
            parts.append(gr)
        ref_g = {k: (parts[0][k] + parts[1][k]) / 2 for k in names}
    # flat gradient buffer holds the SUM over ranks (the 1/world scale is applied inside Adam)
    off = 0
    for k in names:
        n = g.params[k].numel()
        got = flat_grads[off: off + n].view(g.params[k].shape) / 2
        off += (n + 3) & ~3
        ref = ref_g[k]
        assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-7, k
    if exact:
        for k in names:
            ref_upd = g.after(k) - g.params[k]
            upd = sd[k] - g.params[k]
            big = g.grad(k).abs() > 1e-6
            if big.any():
                assert (upd[big] - ref_upd[big]).abs().max().item() <= 1e-3 * g.lr, k


# ---- achieved errors per fixture (recorded for DESIGN.md) ------------------------------------------
def test_record_achieved_errors(device):
    """Not a tolerance test: measures the achieved max |error| of every fixture against the
    reference's goldens and writes them to gpurun_out/r06_achieved_errors.json (with the
    sampled-index flips of the oracle-only cases that ran before it in this session)."""
    rec = {}
    for tag in ("g1_conftest", "g2_mnist_c1", "g3_mnist_ckpt", "g4_resisc_b2"):
        g = Golden(tag)
        eng = _engine(g.cfg, device, g.na, g.nb, g.ns, g.img.shape[1:], g.params)
        i = g.inp
        args = [t.to(device) for t in (i.pos0, i.h0, i.c0, i.hc0, i.cc0, i.q)]
        out = eng.episode_forward(g.img.to(device), *args, g.ref("step_actions").to(device), True)
        e = {"preds_abs": _maxerr(out.step_preds, g.ref("step_preds")),
             "logp_abs": _maxerr(out.step_log_probas, g.ref("step_log_probas")),
             "values_abs": _maxerr(out.step_values, g.ref("step_values")),
             "max_abs_logit": g.ref("step_preds").abs().max().item(),
             "pos_equal": bool(th.equal(out.step_pos.cpu(), g.ref("step_pos")))}
        gp, gl, gv, sc, _ = eng.a2c_loss(out, g.y.to(device), g.gamma)
        e["loss_abs"] = (sc.cpu() - g.ref("loss")).abs().max().item()
        grads = {k: th.zeros_like(v, device=device) for k, v in g.params.items()}
        eng.episode_backward(gp, gl, gv, grads)
        if g.has_full_grads:
            rel = [(_maxerr(grads[k], g.grad(k)) / (g.grad(k).abs().max().item() + 1e-30))
                   for k in g.params if g.grad(k).abs().max().item() > 1e-12]
            e["grad_rel_max"] = max(rel)
            names = list(g.params)
            flat_p = th.cat([g.params[k].flatten() for k in names]).to(device)
            flat_g = th.cat([grads[k].flatten() for k in names])
            m, v = th.zeros_like(flat_p), th.zeros_like(flat_p)
            eng.adam(flat_p, flat_g, m, v, 1, g.lr)
            ref_after = th.cat([g.after(k).flatten() for k in names])
            ref_before = th.cat([g.params[k].flatten() for k in names])
            big = th.cat([g.grad(k).flatten() for k in names]).abs() > 1e-6
            upd, ref_upd = flat_p.cpu() - ref_before, ref_after - ref_before
            e["adam_update_err_over_lr"] = ((upd[big] - ref_upd[big]).abs().max() / g.lr).item()
        rec[tag] = e
    rec["oracle_only_free_running"] = {"cases": len(FLIPS), "samples": sum(f["numel"] for f in FLIPS),
                                       "flips": sum(f["flips"] for f in FLIPS),
                                       "budget": sum(f["allowed"] for f in FLIPS)}
    # margin to every tolerance (tolerance / achieved: > 1 passes; VERDICT r4: the thinnest one is tracked per round)
    tol = {"preds_abs": 1e-5, "logp_abs": 1e-5, "values_abs": 1e-5, "grad_rel_max": 1e-4, "adam_update_err_over_lr": 1e-3}
    for tag, e in rec.items():
        if "preds_abs" in e:
            e["margin"] = {k: (tol[k] / e[k] if e[k] > 0 else float("inf")) for k in tol if k in e}
    rec["thinnest_margin"] = min(m for e in rec.values() if "margin" in e for m in e["margin"].values())
    from tests.util import record

    record("achieved_errors", rec)
    print(json.dumps(rec))
    assert all(r["pos_equal"] for r in rec.values() if isinstance(r, dict) and "pos_equal" in r)
