"""CPU: host-side logic that needs no GPU - the C ABI library loads and exports every symbol
include/marl_hip.h declares, workspace / parameter bookkeeping, state-dict compatibility
with the reference, loud failure on CPU tensors."""
import ctypes as C
import os
import re

import pytest
import torch as th

from oracle import marl_oracle as mo
from tests.util import CASES, Golden, model_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from marlclassification_amd import _lib

    header = open(os.path.join(ROOT, "include", "marl_hip.h")).read()
    declared = set(re.findall(r"\b(marl_[a-z0-9_]+)\s*\(", header))
    lib = _lib.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.marl_abi_version() == _lib.MARL_ABI_VERSION


@pytest.mark.parametrize("tag", list(CASES))
def test_param_table_matches_reference_shapes(tag):
    from marlclassification_amd import _lib
    from marlclassification_amd.engine import state_dict_slots

    cfg = CASES[tag]
    spec = model_spec(cfg)
    mc = spec.config(3, 4, 2, 3, 40, 40)
    lib = _lib.load()
    slots = state_dict_slots(spec.n_cnn_layers)
    shapes = mo.param_shapes(cfg)
    assert set(slots) == set(shapes)
    total = 0
    for name, shape in shapes.items():
        n = lib.marl_param_numel(C.byref(mc), slots[name])
        assert n == int(th.tensor(shape).prod()), name
        total += n
    used = sum(lib.marl_param_numel(C.byref(mc), i) for i in range(_lib.MARL_NPARAMS))
    assert used == total


def test_weights_workspace_layout_is_independent_of_the_batch():
    """the weights workspace (size and the offset of every packed copy inside it) is a function
    of the model and the knobs only.  Batch 256 (R = 4096: image GEMMs on), 255 (R % 32 != 0: off) and 1, 3 agents
    instead of 16, another image size and step count: same bytes, same offsets - while the kernel family does change
    (marl_plan_query).  (Host-side layout code only: runs without a GPU.)"""
    from marlclassification_amd import _lib

    lib = _lib.load()
    spec = model_spec(CASES["g4_resisc_b2"])
    seen = {}
    for na, nb, ns, hw in ((16, 256, 16, 256), (16, 255, 16, 256), (16, 1, 16, 256), (3, 7, 2, 64)):
        mc = spec.config(na, nb, ns, 3, hw, hw)
        wb = C.c_size_t()
        assert lib.marl_workspace_sizes(C.byref(mc), 1, C.byref(wb), None) == 0
        offs = []
        for name in ("LB_WIH", "LA_WHH", "POL_W0", "PRE_W1"):
            off, ld = C.c_int64(0), C.c_int(0)
            assert lib.marl_debug_buffer(C.byref(mc), 1, f"WP{_lib.P[name]}".encode(), 0, C.byref(off), C.byref(ld)) == 0
            offs.append((off.value, ld.value))
        g3, g3m = C.c_int(-1), C.c_int(-1)
        assert lib.marl_plan_query(C.byref(mc), 1, b"g3", C.byref(g3)) == 0
        assert lib.marl_plan_query(C.byref(mc), 1, b"g3_model", C.byref(g3m)) == 0
        seen[(na, nb)] = (wb.value, tuple(offs), g3.value, g3m.value)
    sizes = {v[0] for v in seen.values()}
    layouts = {v[1] for v in seen.values()}
    assert len(sizes) == 1 and len(layouts) == 1, seen
    assert seen[(16, 256)][2] == 1 and seen[(16, 255)][2] == 0 and seen[(16, 1)][2] == 0  # R % 32 decides the kernels
    assert all(v[3] == 1 for v in seen.values())  # ... never the weights workspace
    assert lib.marl_plan_query(C.byref(spec.config(16, 4, 2, 3, 64, 64)), 1, b"no_such_key", C.byref(C.c_int())) != 0


def test_workspace_sizes_and_config_validation():
    from marlclassification_amd import _lib

    lib = _lib.load()
    spec = model_spec(CASES["g4_resisc_b2"])
    mc = spec.config(16, 256, 16, 3, 256, 256)
    wb, eb, eb0 = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert lib.marl_workspace_sizes(C.byref(mc), 1, C.byref(wb), C.byref(eb)) == 0
    assert lib.marl_workspace_sizes(C.byref(mc), 0, None, C.byref(eb0)) == 0
    assert eb.value > eb0.value > 0 and wb.value > 4 * 1686242
    assert sum(lib.marl_param_numel(C.byref(mc), i) for i in range(_lib.MARL_NPARAMS)) == 1686242
    bad = spec.config(16, 256, 16, 3, 8, 8)  # image smaller than the window
    assert lib.marl_workspace_sizes(C.byref(bad), 1, C.byref(wb), C.byref(eb)) < 0
    assert b"bad episode shape" in lib.marl_last_error()


def test_state_dict_is_reference_compatible():
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import MnistCnn, Resisc45Cnn

    g = Golden("g3_mnist_ckpt")  # tensors of the reference's shipped MNIST checkpoint
    m = ModelsWrapper(MnistCnn(6), 80, 80, 16, 24, 8, 2, 5, 10, 112, 112)
    assert list(m.state_dict()) == list(g.params)
    m.load_state_dict(g.params, strict=True)
    for k, v in m.state_dict().items():
        assert th.equal(v, g.params[k])
    r = ModelsWrapper(Resisc45Cnn(12), 256, 256, 64, 96, 16, 2, 4, 45, 384, 384)
    assert sum(p.numel() for p in r.parameters()) == 1686242
    assert r._ModelsWrapper__map_obs.out_size == 256


def test_flat_state_views_and_init():
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import MnistCnn

    th.manual_seed(0)
    m = ModelsWrapper(MnistCnn(12), 23, 22, 21, 20, 19, 2, 4, 10, 24, 25)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = m.flat_state()
    for k, p in m.named_parameters():
        assert p.data_ptr() == flat.param_views()[k].data_ptr()
        assert th.equal(p.data, before[k])
    # reference init recipe: zero biases, unit norm scales, orthogonal(sqrt 2) matrices
    sd = m.state_dict()
    w = sd["_ModelsWrapper__policy.0.weight"]
    assert th.allclose(w.t() @ w, 2 * th.eye(w.shape[1]), atol=1e-5)
    assert sd["_ModelsWrapper__policy.0.bias"].abs().max() == 0
    assert th.equal(sd["_ModelsWrapper__policy.1.weight"], th.ones(25))


INIT_CASES = {
    "mnist_conftest": ("mnist", 12, (23, 22, 21, 20, 19), 4, 10, 24, 25),
    "resisc45_readme": ("resisc45", 12, (256, 256, 64, 96, 16), 4, 45, 384, 384),
    "aid_readme": ("aid", 24, (256, 256, 64, 96, 16), 4, 30, 320, 320),
}


@pytest.mark.parametrize("tag", list(INIT_CASES))
def test_initial_weights_equal_the_reference_under_the_same_seed(tag):
    """``ModelsWrapper(...)`` after ``th.manual_seed(s)`` leaves exactly the
    reference's initial weights (same init recipe, same module order, same RNG draws) - pinned
    by tests/golden/g6_init.npz, generated from the real reference by oracle/make_golden.py."""
    import numpy as np

    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import CNN_BY_NAME

    z = np.load(os.path.join(ROOT, "tests", "golden", "g6_init.npz"))
    ft, f, (n_b, n_a, n_m, n_mo, n_d), n_act, n_cls, nlb, nla = INIT_CASES[tag]
    th.manual_seed(int(z[f"{tag}/seed"][0]))
    m = ModelsWrapper(CNN_BY_NAME[ft](f), n_b, n_a, n_m, n_mo, n_d, 2, n_act, n_cls, nlb, nla)
    sd = m.state_dict()
    assert list(sd) == [str(k) for k in z[f"{tag}/names"]]
    for i, (k, v) in enumerate(sd.items()):
        assert v.double().sum().item() == z[f"{tag}/sum"][i], k
        assert v.double().abs().sum().item() == z[f"{tag}/abs"][i], k
        n = min(8, v.numel())
        assert np.array_equal(v.flatten()[:n].numpy(), z[f"{tag}/head"][i][:n]), k


def test_cpu_tensors_fail_loudly():
    from marlclassification_amd.core import Environment, EpisodeSampler, MultiAgent
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import MnistCnn

    m = ModelsWrapper(MnistCnn(12), 23, 22, 21, 20, 19, 2, 4, 10, 24, 25)
    env = Environment([[1, 0], [-1, 0], [0, 1], [0, -1]], 12)
    sampler = EpisodeSampler(MultiAgent(5, m), env, 7)
    with pytest.raises(RuntimeError, match="GPU"):
        sampler.run_episode(th.rand(19, 1, 28, 28))
    with pytest.raises(RuntimeError, match="GPU"):
        env.reset(th.rand(19, 1, 28, 28), 5)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "marlclassification_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, f)
