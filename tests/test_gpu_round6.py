"""GPU, round 6: (a) a direct HipEngine user survives a knob change between pack() and the compute calls (ADVICE r5,
medium).  Everything goes through the C ABI."""
import pytest
import torch as th

from tests.test_gpu_round5 import _run
from tests.util import Golden, model_spec

pytestmark = pytest.mark.gpu


def test_direct_engine_user_survives_a_knob_change_after_pack(device):
    """`eng.pack(...)` -> `engine.tune(...)` -> `eng.episode_forward(...)` used to hand the
    kernels a freshly ZEROED weights workspace (the tune epoch drops the buffer; only FusedA2C / ModelsWrapper
    re-packed).  The engine now remembers what it packed and packs again by itself; an engine that never packed
    fails loudly instead of running on zeros."""
    from marlclassification_amd import engine as E

    g = Golden("g4_resisc_b2")
    params = {k: v.to(device) for k, v in g.params.items()}
    eng = E.HipEngine(model_spec(g.cfg), device)
    eng.configure(g.na, g.nb, g.ns, g.img.shape[1:])
    eng.pack(params)
    ref_out, ref_grads = _run(eng, g, device, g.nb)
    try:
        E.tune("g3_safe", 0)  # a non-layout knob: same value, new tune epoch -> the workspace is dropped
        gen = eng.weights_generation
        out, grads = _run(eng, g, device, g.nb)
        assert eng.weights_generation == gen + 1  # (it WAS re-allocated, and re-packed behind the caller's back)
        for name in ("step_pos", "step_actions", "step_preds", "step_log_probas", "step_values"):
            assert th.equal(getattr(out, name), getattr(ref_out, name)), name
        for k in grads:
            assert th.equal(grads[k], ref_grads[k]), k
        E.tune("g3", 0)  # a layout knob between forward and backward-less calls: other kernels, same numbers
        out2, _ = _run(eng, g, device, g.nb)
        assert th.equal(out2.step_pos, ref_out.step_pos)
        assert (out2.step_preds - ref_out.step_preds).abs().max().item() <= 2e-5
    finally:
        E.tune("g3", 1)
    fresh = E.HipEngine(model_spec(g.cfg), device)
    fresh.configure(g.na, g.nb, g.ns, g.img.shape[1:])
    with pytest.raises(RuntimeError, match="no weights packed"):
        _run(fresh, g, device, g.nb)
