"""CPU oracle for the MARLClassification hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain restatement, in PyTorch-CPU tensor arithmetic, of the
reference's multi-agent episode rollout and A2C update.  It exists so the HIP
path can be checked against something that is (a) deterministic - every random
draw the reference makes is an *input* here - and (b) pinned against the real
reference: ``oracle/make_golden.py`` imports ``/root/reference`` in the build
container, runs both on the same seeds and requires bit-identical positions,
logits, log-probabilities, values, loss and gradients before it writes the
fixtures under ``tests/golden/`` (parity is therefore PINNED, see the header of
that script and DESIGN.md section "Oracle").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``marlclassification_amd/`` does: the
product path fails loudly when the HIP library is missing instead of falling
back to this code.

Every function cites the reference lines it restates (paths relative to
``/root/reference/marl_classification``).

Row convention: tensors are kept in the reference's ``[Na, Nb, ...]`` layout;
flattened rows are ``r = a * Nb + b`` (``networks/models.py:93``,
``networks/recurrent.py:24-28``, ``core/agent.py:54``).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch as th
import torch.nn.functional as F

Params = Dict[str, th.Tensor]

_P = "_ModelsWrapper__"
_CNN = _P + "map_obs._Generic2dCnnModule__layers."
_LSTM = "._LSTMCellWrapper__lstm."

# CNN stacks of networks/vision.py:55-86,123-127: (in, out) channels + GroupNorm groups
CNN_SPECS = {
    "mnist": ([(1, 8), (8, 16)], [2, 4]),
    "resisc45": ([(3, 16), (16, 32), (32, 64)], [2, 4, 8]),
    "aid": ([(3, 16), (16, 32), (32, 64), (64, 128)], [2, 4, 8, 16]),
    "worldstrat": (
        [(3, 16), (16, 32), (32, 64), (64, 128), (128, 256)],
        [2, 4, 8, 16, 32],
    ),
    "skin_cancer": ([(3, 16), (16, 32), (32, 64)], [2, 4, 8]),
}


@dataclass
class OracleConfig:
    """Shape description of one ModelsWrapper + Environment pair
    (networks/models.py:37-50, core/environment.py:14-16)."""

    ft_extr: str
    window: int
    n_b: int
    n_a: int
    n_m: int
    n_m_o: int
    n_d: int
    nb_class: int
    nlb: int
    nla: int
    actions: List[List[int]] = field(
        default_factory=lambda: [[1, 0], [-1, 0], [0, 1], [0, -1]]
    )
    d: int = 2

    @property
    def cnn_layers(self) -> List[Tuple[int, int]]:
        return CNN_SPECS[self.ft_extr][0]

    @property
    def cnn_groups(self) -> List[int]:
        return CNN_SPECS[self.ft_extr][1]

    @property
    def nb_action(self) -> int:
        return len(self.actions)

    @property
    def cnn_out_hw(self) -> int:
        w = self.window
        for _ in self.cnn_layers:
            w = (w - 3 + 2) // 2 + 1  # networks/vision.py:40-42
        return w

    @property
    def nf(self) -> int:
        return self.cnn_layers[-1][1] * self.cnn_out_hw**2  # vision.py:44

    @property
    def nin(self) -> int:
        return self.nf + self.n_d + self.n_m_o  # models.py:64-69


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """State-dict keys (name-mangled) and shapes, in the
    reference's registration order (networks/models.py:57-74)."""
    s: Dict[str, Tuple[int, ...]] = {}
    for li, (ci, co) in enumerate(cfg.cnn_layers):
        s[f"{_CNN}{3 * li}.weight"] = (co, ci, 3, 3)
        s[f"{_CNN}{3 * li}.bias"] = (co,)
        s[f"{_CNN}{3 * li + 1}.weight"] = (co,)
        s[f"{_CNN}{3 * li + 1}.bias"] = (co,)

    def mlp(prefix: str, dims: Sequence[int], ln_last: bool) -> None:
        # Linear at 0, LayerNorm at 1, SiLU at 2, Linear at 3, [LayerNorm at 4]
        s[f"{_P}{prefix}.0.weight"] = (dims[1], dims[0])
        s[f"{_P}{prefix}.0.bias"] = (dims[1],)
        s[f"{_P}{prefix}.1.weight"] = (dims[1],)
        s[f"{_P}{prefix}.1.bias"] = (dims[1],)
        if len(dims) > 2:
            s[f"{_P}{prefix}.3.weight"] = (dims[2], dims[1])
            s[f"{_P}{prefix}.3.bias"] = (dims[2],)
            if ln_last:
                s[f"{_P}{prefix}.4.weight"] = (dims[2],)
                s[f"{_P}{prefix}.4.bias"] = (dims[2],)

    mlp("map_pos", [cfg.d, cfg.n_d], False)
    mlp("encode_msg", [cfg.n_b, 2 * cfg.n_m, cfg.n_m], True)
    mlp("decode_msg", [cfg.n_m, 2 * cfg.n_m, cfg.n_m_o], True)
    for unit, n in (("belief_unit", cfg.n_b), ("action_unit", cfg.n_a)):
        s[f"{_P}{unit}{_LSTM}weight_ih"] = (4 * n, cfg.nin)
        s[f"{_P}{unit}{_LSTM}weight_hh"] = (4 * n, n)
        s[f"{_P}{unit}{_LSTM}bias_ih"] = (4 * n,)
        s[f"{_P}{unit}{_LSTM}bias_hh"] = (4 * n,)
    mlp("policy", [cfg.n_a, cfg.nla, cfg.nb_action], False)
    mlp("critic", [cfg.n_a, cfg.nla, 1], False)
    mlp("predict", [cfg.n_b, cfg.nlb, cfg.nb_class], False)
    return s


def init_params(cfg: OracleConfig, seed: int) -> Params:
    """networks/init.py:6-29 - orthogonal(gain sqrt 2) weights, zero biases,
    unit/zero norm affines.  (Not draw-for-draw identical to constructing the
    reference module: the reference also consumes the generator for the default
    nn.Module inits before ``apply(init_layers)``.)"""
    g = th.Generator().manual_seed(seed)
    out: Params = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) >= 2:
            w = th.empty(shape)
            th.nn.init.orthogonal_(w, gain=math.sqrt(2.0), generator=g)
            out[name] = w
        elif name.endswith("weight") and not name.endswith(("weight_ih", "weight_hh")):
            out[name] = th.ones(shape)
        else:
            out[name] = th.zeros(shape)
    return out


# --------------------------------------------------------------------------
# environment (core/environment.py)
# --------------------------------------------------------------------------
def crop_patches(img: th.Tensor, pos: th.Tensor, f: int) -> th.Tensor:
    """core/environment.py:95-126 - what the mask + masked_select computes:
    ``obs[a, b] = img[b, :, p0:p0+f, p1:p1+f]`` (dim 0 of pos <-> H)."""
    na, nb, _ = pos.shape
    ar = th.arange(f)
    rows = (pos[..., 0, None] + ar)[..., :, None]  # [Na,Nb,f,1]
    cols = (pos[..., 1, None] + ar)[..., None, :]  # [Na,Nb,1,f]
    bidx = th.arange(nb).view(1, nb, 1, 1)
    # img[b, :, r, c] -> [Na,Nb,f,f,C] -> [Na,Nb,C,f,f]
    return img[bidx, :, rows, cols].permute(0, 1, 4, 2, 3).contiguous()


def crop_patches_masked(img: th.Tensor, pos: th.Tensor, f: int) -> th.Tensor:
    """Same result through the reference's own O(H*W) op sequence
    (core/environment.py:104-126): per-dimension range masks, AND, one
    ``masked_select`` over the broadcast ``[Na, Nb, C, H, W]`` view.  This is
    the "cpu-faithful" baseline variant of BASELINE.md section 3."""
    nb, c, h, w = img.shape
    na = pos.shape[0]
    masks = []
    for d, s in enumerate((h, w)):
        values = th.arange(0, s)
        m = (pos[:, :, d, None] <= values.view(1, 1, s)) & (
            values.view(1, 1, s) < pos[:, :, d, None] + f
        )
        m = m.unsqueeze(-1) if d == 0 else m.unsqueeze(-2)
        masks.append(m)
    mask = (masks[0] & masks[1]).unsqueeze(2)
    return img.unsqueeze(0).masked_select(mask).view(na, nb, c, f, f)


def transition(
    pos: th.Tensor, actions_idx: th.Tensor, table: th.Tensor, f: int, sizes: Sequence[int]
) -> th.Tensor:
    """core/environment.py:56-66,128-150 - the whole move is refused when any
    dimension would leave ``[0, size - f)``; done in fp32 then ``.long()``."""
    mv = table[actions_idx].to(th.float)
    p = pos.to(th.float)
    ok = th.ones(pos.shape[:2], dtype=th.bool)
    for d in range(pos.shape[-1]):
        ok = ok & (p[..., d] + mv[..., d] >= 0) & (p[..., d] + mv[..., d] + f < sizes[d])
    okf = ok.unsqueeze(2).to(th.float)
    return (okf * (p + mv) + (1 - okf) * p).to(th.long)


def normalized_positions(pos: th.Tensor, sizes: Sequence[int]) -> th.Tensor:
    """core/environment.py:74-81."""
    return pos.to(th.float) / th.tensor([[list(sizes)]], dtype=th.float)


# --------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------
def cnn_forward(p: Params, cfg: OracleConfig, patches: th.Tensor) -> th.Tensor:
    """networks/vision.py:23-52 (+ :63-65 for MNIST's channel-0 slice):
    L x [Conv2d k3 s2 p1 -> GroupNorm -> SiLU] -> Flatten(C,h,w)."""
    x = patches
    if cfg.ft_extr == "mnist":
        x = x[:, 0, None, :, :]
    for li, g in enumerate(cfg.cnn_groups):
        x = F.conv2d(
            x, p[f"{_CNN}{3 * li}.weight"], p[f"{_CNN}{3 * li}.bias"], stride=2, padding=1
        )
        x = F.group_norm(
            x, g, p[f"{_CNN}{3 * li + 1}.weight"], p[f"{_CNN}{3 * li + 1}.bias"], 1e-5
        )
        x = F.silu(x)
    return x.flatten(1, -1)


def aggregate_messages(m: th.Tensor) -> th.Tensor:
    """networks/message.py:5-17."""
    na = m.shape[0]
    if na == 1:
        return th.zeros_like(m)
    return (m.sum(dim=0) - m) / (na - 1)


def _lin(p: Params, prefix: str, i: int, x: th.Tensor) -> th.Tensor:
    return F.linear(x, p[f"{_P}{prefix}.{i}.weight"], p[f"{_P}{prefix}.{i}.bias"])


def _ln_silu(p: Params, prefix: str, i: int, x: th.Tensor) -> th.Tensor:
    w = p[f"{_P}{prefix}.{i}.weight"]
    return F.silu(F.layer_norm(x, (w.shape[0],), w, p[f"{_P}{prefix}.{i}.bias"], 1e-5))


def mlp2_ln(p: Params, prefix: str, x: th.Tensor) -> th.Tensor:
    """MessageSender / MessageReceiver (networks/message.py:20-49):
    Linear-LN-SiLU-Linear-LN-SiLU."""
    x = _ln_silu(p, prefix, 1, _lin(p, prefix, 0, x))
    return _ln_silu(p, prefix, 4, _lin(p, prefix, 3, x))


def head(p: Params, prefix: str, x: th.Tensor) -> th.Tensor:
    """Policy / Critic / Prediction trunk (networks/policy.py:12-15,23-26,
    networks/prediction.py:11-14): Linear-LN-SiLU-Linear."""
    return _lin(p, prefix, 3, _ln_silu(p, prefix, 1, _lin(p, prefix, 0, x)))


def lstm_cell(
    p: Params, unit: str, u: th.Tensor, h: th.Tensor, c: th.Tensor
) -> Tuple[th.Tensor, th.Tensor, th.Tensor]:
    """networks/recurrent.py:19-35 -> nn.LSTMCell: gate order i,f,g,o."""
    k = f"{_P}{unit}{_LSTM}"
    gates = F.linear(u, p[k + "weight_ih"], p[k + "bias_ih"]) + F.linear(
        h, p[k + "weight_hh"], p[k + "bias_hh"]
    )
    i, f_, g, o = gates.chunk(4, dim=-1)
    i, f_, g, o = th.sigmoid(i), th.sigmoid(f_), th.tanh(g), th.sigmoid(o)
    c2 = f_ * c + i * g
    h2 = o * th.tanh(c2)
    return h2, c2, th.cat((i, f_, g, o), dim=-1)


@dataclass
class StepOut:
    probs: th.Tensor  # [Na,Nb,nA]
    values: th.Tensor  # [Na,Nb]
    preds: th.Tensor  # [Na,Nb,nC]
    msg: th.Tensor  # [Na,Nb,n_m]
    h: th.Tensor
    c: th.Tensor
    hc: th.Tensor
    cc: th.Tensor
    u: th.Tensor  # [Na,Nb,nin]


def step_forward(
    p: Params,
    cfg: OracleConfig,
    obs: th.Tensor,
    msg: th.Tensor,
    norm_pos: th.Tensor,
    h: th.Tensor,
    c: th.Tensor,
    hc: th.Tensor,
    cc: th.Tensor,
) -> StepOut:
    """ModelsWrapper.forward, networks/models.py:78-138."""
    na, nb = obs.shape[:2]
    b_t = cnn_forward(p, cfg, obs.flatten(0, 1)).view(na, nb, -1)  # :92-94
    d_bar = mlp2_ln(p, "decode_msg", aggregate_messages(msg))  # :97-98
    lam = _ln_silu(p, "map_pos", 1, _lin(p, "map_pos", 0, norm_pos))  # :101
    u = th.cat((b_t, d_bar, lam), dim=2)  # :104
    uf = u.flatten(0, 1)
    h2, c2, _ = lstm_cell(p, "belief_unit", uf, h.flatten(0, 1), c.flatten(0, 1))  # :107
    h2, c2 = h2.view(na, nb, -1), c2.view(na, nb, -1)
    new_msg = mlp2_ln(p, "encode_msg", h2)  # :114-116
    hc2, cc2, _ = lstm_cell(p, "action_unit", uf, hc.flatten(0, 1), cc.flatten(0, 1))  # :119
    hc2, cc2 = hc2.view(na, nb, -1), cc2.view(na, nb, -1)
    probs = th.softmax(head(p, "policy", hc2), dim=-1)  # :126-128, policy.py:16
    values = head(p, "critic", hc2).flatten(-2, -1)  # :131, policy.py:27
    preds = head(p, "predict", h2)  # :134
    return StepOut(probs, values, preds, new_msg, h2, c2, hc2, cc2, u)


def sample_actions(probs: th.Tensor, q: th.Tensor) -> th.Tensor:
    """core/agent.py:53-55 - ``th.multinomial(p, 1)`` is ``argmax(p / q)`` with
    ``q ~ Exp(1)`` (aten multinomial, n_sample == 1 path); ``q`` is injected.
    probs, q: [Na,Nb,nA] -> int64 [Na,Nb]."""
    return th.argmax(probs / q, dim=-1)


# --------------------------------------------------------------------------
# episode (core/episode.py:32-82)
# --------------------------------------------------------------------------
@dataclass
class EpisodeInputs:
    """Every random draw of one reference episode, in the reference's draw
    order: randint(H-f), randint(W-f), randn h, c, h^, c^,
    then one Exp(1) tensor [Na*Nb, nA] per step."""

    pos0: th.Tensor  # int64 [Na,Nb,2]
    h0: th.Tensor
    c0: th.Tensor
    hc0: th.Tensor
    cc0: th.Tensor
    q: th.Tensor  # [Ns,Na,Nb,nA]


def draw_episode_inputs(
    cfg: OracleConfig, na: int, nb: int, ns: int, sizes: Sequence[int], seed: int
) -> EpisodeInputs:
    """Consumes the default CPU generator exactly as the reference does between
    ``th.manual_seed(seed)`` and the end of ``run_episode``
    (core/environment.py:33-43, networks/models.py:148-159, core/agent.py:53-55)."""
    th.manual_seed(seed)
    pos0 = th.stack(
        [th.randint(s - cfg.window, (na, nb)) for s in sizes], dim=-1
    )
    h0 = th.randn(na, nb, cfg.n_b)
    c0 = th.randn(na, nb, cfg.n_b)
    hc0 = th.randn(na, nb, cfg.n_a)
    cc0 = th.randn(na, nb, cfg.n_a)
    q = th.stack(
        [th.empty(na * nb, cfg.nb_action).exponential_(1).view(na, nb, -1) for _ in range(ns)]
    )
    return EpisodeInputs(pos0, h0, c0, hc0, cc0, q)


@dataclass
class EpisodeTrace:
    step_preds: th.Tensor  # [Ns,Na,Nb,nC]
    step_log_probas: th.Tensor  # [Ns,Na,Nb]
    step_values: th.Tensor  # [Ns,Na,Nb]
    step_pos: th.Tensor  # int64 [Ns,Na,Nb,2]  (position AFTER move t)
    step_actions: th.Tensor  # int64 [Ns,Na,Nb]
    step_probs: th.Tensor  # [Ns,Na,Nb,nA]
    step_u: th.Tensor  # [Ns,Na,Nb,nin]
    step_h: th.Tensor  # [Ns,Na,Nb,n_b]
    step_hc: th.Tensor  # [Ns,Na,Nb,n_a]
    step_msg: th.Tensor  # [Ns,Na,Nb,n_m]


def run_episode(
    p: Params,
    cfg: OracleConfig,
    img: th.Tensor,
    inp: EpisodeInputs,
    ns: int,
    faithful_crop: bool = False,
    forced_actions: Optional[th.Tensor] = None,
) -> EpisodeTrace:
    """EpisodeSampler.__episode_impl (core/episode.py:32-82) with injected
    randomness.  ``forced_actions`` ([Ns,Na,Nb] int64) teacher-forces the action
    sequence (used at sizes where one ulp in a probability may flip argmax)."""
    crop = crop_patches_masked if faithful_crop else crop_patches
    sizes = list(img.shape[2:])
    table = th.tensor(cfg.actions)
    na, nb = inp.pos0.shape[:2]
    pos = inp.pos0
    h, c, hc, cc = inp.h0, inp.c0, inp.hc0, inp.cc0
    msg = th.zeros(na, nb, cfg.n_m)  # models.py:161-162
    obs = crop(img, pos, cfg.window)  # environment.py:45
    keys = ("preds", "logp", "values", "pos", "act", "probs", "u", "h", "hc", "msg")
    acc: Dict[str, list] = {k: [] for k in keys}
    for t in range(ns):
        so = step_forward(p, cfg, obs, msg, normalized_positions(pos, sizes), h, c, hc, cc)
        h, c, hc, cc, msg = so.h, so.c, so.hc, so.cc, so.msg  # agent.py:48-49
        a = sample_actions(so.probs, inp.q[t]) if forced_actions is None else forced_actions[t]
        logp = th.gather(so.probs, -1, a.unsqueeze(-1)).squeeze(-1).log()  # agent.py:57-61
        pos = transition(pos, a, table, cfg.window, sizes)  # environment.py:56-66
        if t + 1 < ns or faithful_crop:
            obs = crop(img, pos, cfg.window)  # last crop is discarded (episode.py:72)
        for k, v in zip(
            keys, (so.preds, logp, so.values, pos, a, so.probs, so.u, so.h, so.hc, so.msg)
        ):
            acc[k].append(v)
    st = {k: th.stack(v) for k, v in acc.items()}
    return EpisodeTrace(
        st["preds"], st["logp"], st["values"], st["pos"], st["act"],
        st["probs"], st["u"], st["h"], st["hc"], st["msg"],
    )


# --------------------------------------------------------------------------
# A2C update (training/functions.py, training/trainer.py:76-116)
# --------------------------------------------------------------------------
def classification_rewards(step_preds: th.Tensor, targets: th.Tensor) -> th.Tensor:
    """training/functions.py:7-32."""
    ns, na, _, nc = step_preds.shape
    tgt = targets[:, None, None].repeat(1, ns, na)
    err = F.cross_entropy(step_preds.permute(2, 3, 0, 1), tgt, reduction="none").permute(1, 2, 0)
    rnd = math.log(nc)
    return (rnd - err) / rnd


def discounted_returns(rewards: th.Tensor, gamma: float) -> th.Tensor:
    """training/functions.py:35-51 (flip-cumsum-flip form, kept for its fp32
    rounding)."""
    shape = [rewards.size(0)] + [1] * (rewards.dim() - 1)
    t_steps = th.arange(rewards.size(0)).view(*shape).to(th.float)
    ret = rewards * gamma**t_steps
    return ret.flip(dims=(0,)).cumsum(0).flip(dims=(0,)) / gamma**t_steps


def advantages(step_preds: th.Tensor, step_values: th.Tensor, y: th.Tensor, gamma: float) -> th.Tensor:
    """returns - values (training/trainer.py:89-92), before standardisation."""
    return discounted_returns(classification_rewards(step_preds, y), gamma) - step_values


def standardize(values: th.Tensor, eps: float = 1e-8) -> th.Tensor:
    """training/functions.py:54-55 (unbiased std over ALL elements)."""
    return (values - values.mean()) / (values.std() + eps)


@dataclass
class LossOut:
    loss: th.Tensor
    path: th.Tensor  # scalar: path_loss.sum(0).mean()
    error: th.Tensor  # scalar: error.mean()
    critic: th.Tensor  # scalar: critic_loss.sum(0).mean()


def a2c_loss(
    step_preds: th.Tensor,
    step_log_probas: th.Tensor,
    step_values: th.Tensor,
    y: th.Tensor,
    gamma: float,
    adv_stats: Optional[Tuple[float, float]] = None,
) -> LossOut:
    """training/trainer.py:76-111 and the scalars of :119-122.  ``adv_stats`` = (mean, std)
    overrides the statistics ``standardize`` would compute from this batch (used to check
    the data-parallel "exact standardize" exchange against a big-batch run)."""
    ns, _, nb, _ = step_preds.shape
    predictions = step_preds.mean(dim=1).flatten(0, 1)
    targets = y.unsqueeze(0).repeat(ns, 1).flatten(0, 1)
    error = F.cross_entropy(predictions, targets, reduction="none").unflatten(0, (ns, 1, nb))
    rewards = classification_rewards(step_preds, y)
    returns = discounted_returns(rewards, gamma)
    if adv_stats is None:
        adv = standardize(returns - step_values)
    else:
        adv = ((returns - step_values) - adv_stats[0]) / (adv_stats[1] + 1e-8)
    path = -step_log_probas * adv.detach()
    actor = path + error
    critic = F.smooth_l1_loss(step_values, returns.detach(), reduction="none")
    loss = th.sum(actor + critic, 0).mean()
    return LossOut(loss, path.sum(dim=0).mean(), error.mean(), critic.sum(dim=0).mean())


def adam_step(
    params: Params,
    grads: Params,
    m: Params,
    v: Params,
    step: int,
    lr: float,
    betas: Tuple[float, float] = (0.9, 0.999),
    eps: float = 1e-8,
) -> None:
    """th.optim.Adam defaults as used at training/trainer.py:33 (no weight
    decay, no amsgrad), single-tensor formulation; in place; ``step`` is the
    1-based count of this update."""
    b1, b2 = betas
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    for k, w in params.items():
        g = grads[k]
        m[k].lerp_(g, 1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(m[k], denom, value=-(lr / bc1))


def train_iteration(
    p: Params,
    cfg: OracleConfig,
    img: th.Tensor,
    y: th.Tensor,
    inp: EpisodeInputs,
    ns: int,
    gamma: float,
    faithful_crop: bool = False,
    forced_actions: Optional[th.Tensor] = None,
    adv_stats: Optional[Tuple[float, float]] = None,
) -> Tuple[EpisodeTrace, LossOut, Params]:
    """One iteration of Trainer.train_epoch's body (training/trainer.py:73-114)
    up to and including ``loss.backward()``: returns the trace, the loss
    scalars and every parameter gradient (autograd on this restatement)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    tr = run_episode(leaves, cfg, img, inp, ns, faithful_crop, forced_actions)
    lo = a2c_loss(tr.step_preds, tr.step_log_probas, tr.step_values, y, gamma, adv_stats)
    lo.loss.backward()
    grads = {
        k: (v.grad if v.grad is not None else th.zeros_like(v)) for k, v in leaves.items()
    }
    return tr, lo, grads
