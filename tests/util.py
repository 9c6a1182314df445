"""Helpers shared by the tests: golden-fixture loading (tests/golden/*.npz, written by
oracle/make_golden.py from the real reference) and the oracle <-> engine glue."""
import os

import numpy as np
import torch as th

from oracle import marl_oracle as mo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r06"  # prefix of the parity records the GPU tests write (copied into profiles/ after the box run)


def record(name, obj):
    """Achieved-error / margin records of the GPU parity tests -> gpurun_out/<ROUND>_<name>.json (gpurun_out/ is
    what comes back from the GPU box; tools/round_profiles.sh copies these into profiles/)."""
    import json

    path = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, f"{ROUND}_{name}.json"), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass

CASES = {
    "g1_conftest": mo.OracleConfig("mnist", 12, 23, 22, 21, 20, 19, 10, 24, 25),
    "g2_mnist_c1": mo.OracleConfig("mnist", 6, 64, 64, 16, 24, 8, 10, 96, 96),
    "g3_mnist_ckpt": mo.OracleConfig(
        "mnist", 6, 80, 80, 16, 24, 8, 10, 112, 112,
        actions=[[1, 0], [-1, 0], [0, 1], [0, -1], [0, 0]],
    ),
    "g4_resisc_b2": mo.OracleConfig("resisc45", 12, 256, 256, 64, 96, 16, 45, 384, 384),
}


def uniform_params(cfg, seed):
    """Same machine-independent init as oracle/make_golden.py::uniform_params."""
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


class Golden:
    def __init__(self, tag):
        self.tag = tag
        self.cfg = CASES[tag]
        z = np.load(os.path.join(GOLDEN, tag + ".npz"))
        self.z = z
        self.na, self.nb, self.ns, self.seed = (int(v) for v in z["meta_na_nb_ns_seed"])
        self.lr, self.gamma = (float(v) for v in z["meta_lr_gamma"])
        if "img" in z:
            self.img = th.from_numpy(z["img"])
        else:
            s = [int(v) for v in z["img_seed_shape"]]
            self.img = th.rand(*s[1:], generator=th.Generator().manual_seed(s[0]))
        self.y = th.from_numpy(z["y"])
        self.inp = mo.EpisodeInputs(*(th.from_numpy(z[k]) for k in ("pos0", "h0", "c0", "hc0", "cc0", "q")))
        if any(k.startswith("param/") for k in z.files):
            self.params = {k: th.from_numpy(z["param/" + k]) for k in mo.param_shapes(self.cfg)}
        else:
            self.params = uniform_params(self.cfg, int(z["params_uniform_seed"][0]))
        chk = sum(v.double().sum().item() for v in self.params.values())
        assert abs(chk - float(z["params_checksum"][0])) < 1e-6 * max(1.0, abs(chk)), "fixture params drifted"
        self.has_full_grads = any(k.startswith("grad/") for k in z.files)

    def ref(self, key):
        return th.from_numpy(self.z["ref_" + key])

    def grad(self, name):
        return th.from_numpy(self.z["grad/" + name])

    def after(self, name):
        return th.from_numpy(self.z["after/" + name])


def model_spec(cfg):
    from marlclassification_amd.engine import ModelSpec

    return ModelSpec(cfg.ft_extr, cfg.window, cfg.n_b, cfg.n_a, cfg.n_m, cfg.n_m_o, cfg.n_d,
                     cfg.nb_class, cfg.nlb, cfg.nla, [list(a) for a in cfg.actions])
