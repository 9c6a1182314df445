"""CPU: the oracle reproduces every golden vector generated from the real reference
(oracle/make_golden.py).  Bit-exact: same torch build, same op sequence."""
import numpy as np
import pytest
import torch as th

from oracle import marl_oracle as mo
from tests.util import CASES, GOLDEN, Golden


@pytest.mark.parametrize("tag", ["g1_conftest", "g2_mnist_c1", "g3_mnist_ckpt"])
def test_oracle_matches_reference_episode_and_update(tag):
    g = Golden(tag)
    tr, lo, grads = mo.train_iteration(g.params, g.cfg, g.img, g.y, g.inp, g.ns, g.gamma)
    assert th.equal(tr.step_pos, g.ref("step_pos"))
    assert th.equal(tr.step_actions, g.ref("step_actions"))
    for a, b in ((tr.step_preds, "step_preds"), (tr.step_log_probas, "step_log_probas"),
                 (tr.step_values, "step_values")):
        assert th.allclose(a.detach(), g.ref(b), rtol=0, atol=1e-6), b
    assert abs(lo.loss.item() - g.ref("loss")[0].item()) < 1e-4
    for k in g.params:
        assert th.allclose(grads[k], g.grad(k), rtol=1e-4, atol=1e-6), k
    after = {k: v.clone() for k, v in g.params.items()}
    m = {k: th.zeros_like(v) for k, v in after.items()}
    v = {k: th.zeros_like(x) for k, x in after.items()}
    mo.adam_step(after, grads, m, v, 1, g.lr)
    for k in g.params:
        assert th.allclose(after[k], g.after(k), rtol=1e-5, atol=1e-7), k


def test_oracle_resisc_dims_rollout():
    g = Golden("g4_resisc_b2")
    tr = mo.run_episode(g.params, g.cfg, g.img, g.inp, g.ns)
    assert th.equal(tr.step_pos, g.ref("step_pos"))
    assert th.allclose(tr.step_preds, g.ref("step_preds"), rtol=0, atol=2e-6)
    assert th.allclose(tr.step_values, g.ref("step_values"), rtol=0, atol=2e-6)


def test_faithful_crop_equals_gather_crop():
    g = Golden("g1_conftest")
    a = mo.run_episode(g.params, g.cfg, g.img, g.inp, g.ns, faithful_crop=True)
    b = mo.run_episode(g.params, g.cfg, g.img, g.inp, g.ns, faithful_crop=False)
    assert th.equal(a.step_preds, b.step_preds) and th.equal(a.step_pos, b.step_pos)


def test_unit_kats():
    z = np.load(GOLDEN + "/g5_unit_kats.npz")
    t = lambda k: th.from_numpy(z[k])
    assert th.equal(mo.crop_patches(t("crop_img"), t("crop_pos"), 5), t("crop_obs"))
    assert th.equal(mo.crop_patches_masked(t("crop_img"), t("crop_pos"), 5), t("crop_obs"))
    acts = th.arange(t("tr_table").shape[0]).view(-1, 1)
    new = mo.transition(t("tr_pos"), acts, t("tr_table"), 5, [10, 10])
    assert th.equal(new, t("tr_new"))
    # SURVEY 8c border cases from (4,0), img 10, f 5
    assert new[:, 0].tolist() == [[4, 0], [3, 0], [4, 1], [4, 0], [4, 0], [4, 0]]
    assert th.equal(mo.aggregate_messages(t("agg_in")), t("agg_out"))
    assert th.equal(mo.aggregate_messages(t("agg_in")[:1]), th.zeros_like(t("agg_in")[:1]))
    assert th.equal(mo.discounted_returns(t("ret_in"), 0.99), t("ret_out"))
    assert th.equal(mo.standardize(t("ret_in")), t("std_out"))
    assert th.equal(mo.classification_rewards(t("rw_preds"), t("rw_y")), t("rw_out"))
    assert th.equal(mo.sample_actions(t("mn_probs"), t("mn_q")), t("mn_actions"))


def test_param_shapes_match_checkpoint_layout():
    cfg = CASES["g3_mnist_ckpt"]
    g = Golden("g3_mnist_ckpt")
    shapes = mo.param_shapes(cfg)
    assert list(shapes) == list(g.params)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 149616
