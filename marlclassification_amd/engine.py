"""Thin host-side driver over libmarl_hip.so: builds ``marl_config`` structs, owns the two
workspaces the C ABI asks the caller to provide, and turns torch tensors into raw device
pointers + the current HIP stream.  PyTorch is used for device memory, streams and (in
``parallel.py``) ``torch.distributed`` only - every computation happens in the HIP library.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch as th

from . import _lib
from ._lib import MARL_NPARAMS, MarlConfig, P, check

# Conv stacks of the reference's feature extractors (networks/vision.py:55-86,123-127):
# channels entering / leaving every Conv-GroupNorm-SiLU block and the GroupNorm groups.
CNN_SPECS: Dict[str, Tuple[List[int], List[int]]] = {
    "mnist": ([1, 8, 16], [2, 4]),
    "resisc45": ([3, 16, 32, 64], [2, 4, 8]),
    "aid": ([3, 16, 32, 64, 128], [2, 4, 8, 16]),
    "worldstrat": ([3, 16, 32, 64, 128, 256], [2, 4, 8, 16, 32]),
    "skin_cancer": ([3, 16, 32, 64], [2, 4, 8]),
}

_MW = "_ModelsWrapper__"
_CNN_PREFIX = _MW + "map_obs._Generic2dCnnModule__layers."
_LSTM = "._LSTMCellWrapper__lstm."


def _mlp_names(prefix: str, tag: str, two_ln: bool) -> Dict[str, int]:
    out = {
        f"{_MW}{prefix}.0.weight": P[f"{tag}_W0"],
        f"{_MW}{prefix}.0.bias": P[f"{tag}_B0"],
        f"{_MW}{prefix}.1.weight": P[f"{tag}_LN0W" if two_ln else f"{tag}_LNW"],
        f"{_MW}{prefix}.1.bias": P[f"{tag}_LN0B" if two_ln else f"{tag}_LNB"],
        f"{_MW}{prefix}.3.weight": P[f"{tag}_W1"],
        f"{_MW}{prefix}.3.bias": P[f"{tag}_B1"],
    }
    if two_ln:
        out[f"{_MW}{prefix}.4.weight"] = P[f"{tag}_LN1W"]
        out[f"{_MW}{prefix}.4.bias"] = P[f"{tag}_LN1B"]
    return out


def state_dict_slots(n_cnn_layers: int) -> Dict[str, int]:
    """Reference state-dict key (name-mangled) -> slot of the C parameter
    table (enum in include/marl_hip.h)."""
    m: Dict[str, int] = {}
    for l in range(n_cnn_layers):
        m[f"{_CNN_PREFIX}{3 * l}.weight"] = 4 * l
        m[f"{_CNN_PREFIX}{3 * l}.bias"] = 4 * l + 1
        m[f"{_CNN_PREFIX}{3 * l + 1}.weight"] = 4 * l + 2
        m[f"{_CNN_PREFIX}{3 * l + 1}.bias"] = 4 * l + 3
    m[f"{_MW}map_pos.0.weight"] = P["POS_W"]
    m[f"{_MW}map_pos.0.bias"] = P["POS_B"]
    m[f"{_MW}map_pos.1.weight"] = P["POS_LNW"]
    m[f"{_MW}map_pos.1.bias"] = P["POS_LNB"]
    m.update(_mlp_names("encode_msg", "ENC", True))
    m.update(_mlp_names("decode_msg", "DEC", True))
    for unit, tag in (("belief_unit", "LB"), ("action_unit", "LA")):
        m[f"{_MW}{unit}{_LSTM}weight_ih"] = P[f"{tag}_WIH"]
        m[f"{_MW}{unit}{_LSTM}weight_hh"] = P[f"{tag}_WHH"]
        m[f"{_MW}{unit}{_LSTM}bias_ih"] = P[f"{tag}_BIH"]
        m[f"{_MW}{unit}{_LSTM}bias_hh"] = P[f"{tag}_BHH"]
    m.update(_mlp_names("policy", "POL", False))
    m.update(_mlp_names("critic", "CRI", False))
    m.update(_mlp_names("predict", "PRE", False))
    return m


@dataclass
class ModelSpec:
    """Everything ``marl_config`` needs that does not depend on the batch."""

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

    def config(self, nb_agents: int, batch: int, nb_steps: int, img_c: int, h: int, w: int,
               img_u8: bool = False) -> MarlConfig:
        if self.ft_extr not in CNN_SPECS:
            raise ValueError(
                f'feature extractor "{self.ft_extr}" has no HIP implementation '
                f"(supported: {sorted(CNN_SPECS)})"
            )
        ch, groups = CNN_SPECS[self.ft_extr]
        if len(self.actions) > _lib.MARL_MAX_ACTIONS:
            raise ValueError("too many actions")
        cfg = MarlConfig()
        cfg.nb_agents, cfg.batch, cfg.nb_steps = nb_agents, batch, nb_steps
        cfg.img_c, cfg.img_h, cfg.img_w = img_c, h, w
        cfg.window = self.window
        cfg.cnn_layers = len(groups)
        for i, c in enumerate(ch):
            cfg.cnn_ch[i] = c
        for i, g in enumerate(groups):
            cfg.cnn_groups[i] = g
        cfg.n_b, cfg.n_a, cfg.n_m, cfg.n_m_o, cfg.n_d = self.n_b, self.n_a, self.n_m, self.n_m_o, self.n_d
        cfg.nb_action, cfg.nb_class = len(self.actions), self.nb_class
        cfg.nlb, cfg.nla = self.nlb, self.nla
        for j, (d0, d1) in enumerate(self.actions):
            cfg.actions[j][0], cfg.actions[j][1] = d0, d1
        cfg.img_u8 = int(img_u8)
        return cfg

    @property
    def n_cnn_layers(self) -> int:
        return len(CNN_SPECS[self.ft_extr][1])


_U64 = (1 << 64) - 1

# Layout-affecting knobs (marl_tune: tile plans, split-K targets, mfma_split ...) change the sizes
# of regions INSIDE the episode workspace.  The library recomputes the layout on every call but
# never sees the workspace size, so a buffer allocated before a knob changed must not be reused:
# ``tune`` bumps this epoch and every HipEngine drops its cached workspaces when it differs.
_tune_epoch = 0


def tune(key: str, value: int) -> None:
    """marl_tune + invalidation of every engine's cached workspaces (set knobs through this)."""
    global _tune_epoch
    check(_lib.load().marl_tune(key.encode(), int(value)))
    _tune_epoch += 1


def _nbytes(t: th.Tensor) -> int:
    return t.numel() * t.element_size()


def _ptr(t: Optional[th.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: th.device) -> int:
    return th.cuda.current_stream(device).cuda_stream


def _need(t: th.Tensor, dtype: th.dtype, name: str) -> th.Tensor:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} lives on {t.device}: the HIP path only runs on a GPU "
            "(there is no CPU implementation in this package)"
        )
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


@dataclass
class EpisodeTensors:
    step_preds: th.Tensor
    step_log_probas: th.Tensor
    step_values: th.Tensor
    step_pos: th.Tensor
    step_actions: th.Tensor


class HipEngine:
    """Owns workspaces for one model on one device and issues the C ABI calls."""

    def __init__(self, spec: ModelSpec, device: th.device) -> None:
        self.lib = _lib.load()
        self.spec = spec
        self.device = th.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipEngine needs a GPU device (torch 'cuda' == HIP on ROCm)")
        self.slots = state_dict_slots(spec.n_cnn_layers)
        self._wws: Optional[th.Tensor] = None
        self._wws_epoch = _tune_epoch   # tune epoch the weights workspace was laid out under
        self._wws_bytes = 0
        # bumped whenever the weights workspace is (re)allocated: its packed copies are gone and whoever
        # packed last (FusedA2C, ModelsWrapper.ensure_packed) must pack again before the next call
        self.weights_generation = 0
        # what pack() last filled: the generation it packed into and the tensors it packed from (a
        # direct user doing pack -> tune -> forward used to get a freshly ZEROED weights workspace, silently)
        self._packed_generation: Optional[int] = None
        self._packed_params: Optional[Dict[str, th.Tensor]] = None
        self._ews: Dict[Tuple, th.Tensor] = {}
        self._cfg_key: Optional[Tuple] = None
        self.cfg: Optional[MarlConfig] = None
        self._packed_version: Optional[int] = None
        self._tune_seen = _tune_epoch
        # what the last training rollout left behind for backward: the episode workspace holds
        # its activations, `_fwd_img` keeps its image batch alive (the first convolution's weight
        # gradient re-gathers the patches), `fwd_generation` lets callers detect a stale backward
        self.fwd_generation = 0
        self.pack_generation = 0       # bumped by pack(): a live episode's backward needs ITS weights
        self._ws_pool: Dict[tuple, list] = {}  # released per-episode training workspaces (autograd path)
        self._fwd_img: Optional[th.Tensor] = None
        self._fwd_key: Optional[Tuple] = None

    # -- configuration / workspaces -------------------------------------------------
    def configure(self, nb_agents: int, batch: int, nb_steps: int, img_shape: Sequence[int],
                  img_u8: bool = False) -> MarlConfig:
        key = (nb_agents, batch, nb_steps, tuple(img_shape), bool(img_u8))
        if key != self._cfg_key:
            c, h, w = img_shape
            self.cfg = self.spec.config(nb_agents, batch, nb_steps, c, h, w, img_u8)
            self._cfg_key = key
            # The weights layout is a function of the MODEL and the knobs only (csrc/episode.hip,
            # g3_model_ok) - belt and braces: should a configuration ever ask for another size, the packed
            # copies of the old layout must not be read through the new one.
            if self._wws is not None and self._sizes(True)[0] != self._wws_bytes:
                self._wws = None
        assert self.cfg is not None
        return self.cfg

    def _sizes(self, train: bool) -> Tuple[int, int]:
        wb, eb = C.c_size_t(0), C.c_size_t(0)
        check(self.lib.marl_workspace_sizes(C.byref(self.cfg), int(train), C.byref(wb), C.byref(eb)))
        return wb.value, eb.value

    def weights_ws(self) -> th.Tensor:
        if self._wws is not None and self._wws_epoch != _tune_epoch:
            # a knob moved (g3, g3_min_units, mfma_split ... decide which weight images exist): the old
            # buffer has the old layout AND the old contents - drop it, and make every packer pack again
            self._wws = None
        if self._wws is None:
            wb, _ = self._sizes(True)
            self._wws = th.zeros(wb // 4 + 64, dtype=th.float32, device=self.device)
            self._wws_epoch = _tune_epoch
            self._wws_bytes = wb
            self.weights_generation += 1
        return self._wws

    def packed_weights_ws(self) -> th.Tensor:
        """The weights workspace for a compute call: if a knob change dropped the buffer pack() filled, the
        parameters of the last pack() are packed again into the new layout (their CURRENT values - they are the
        caller's live tensors); with nothing ever packed the call fails instead of running on zeros."""
        wws = self.weights_ws()
        if self._packed_generation != self.weights_generation:
            if self._packed_params is None:
                raise RuntimeError("HipEngine: no weights packed - call pack(params) before a compute call")
            self.pack(self._packed_params)
            wws = self.weights_ws()
        return wws

    def weights_token(self) -> int:
        """Generation of the weights workspace AFTER applying pending invalidations (callers compare it
        with the value they saw when they last packed)."""
        self.weights_ws()
        return self.weights_generation

    def episode_ws(self, train: bool) -> th.Tensor:
        if self._tune_seen != _tune_epoch:  # a layout knob changed: sizes may have moved
            self._ews.clear()
            self._fwd_img = None
            self._tune_seen = _tune_epoch
        key = (self._cfg_key, train)
        ws = self._ews.get(key)
        if ws is None:
            _, eb = self._sizes(train)
            # keep at most one workspace per mode: shapes rarely change
            for k in [k for k in self._ews if k[1] == train]:
                del self._ews[k]
            ws = th.zeros(eb // 4 + 64, dtype=th.float32, device=self.device)
            self._ews[key] = ws
        return ws

    def train_ws_acquire(self) -> th.Tensor:
        """A training workspace owned by ONE episode (the autograd path: several rollouts of the same
        model may be alive at once, as with the reference's autograd graph - core/episode.py:84 there).
        Released workspaces of the current configuration are reused; the fused Trainer keeps using
        the engine's own workspace (``episode_ws``)."""
        key = (self._cfg_key, _tune_epoch)
        for k in [k for k in self._ws_pool if k != key]:
            del self._ws_pool[k]  # other shape / layout: free the memory
        free = self._ws_pool.setdefault(key, [])
        if free:
            return free.pop()
        _, eb = self._sizes(True)
        ws = th.zeros(eb // 4 + 64, dtype=th.float32, device=self.device)
        ws._marl_key = key
        return ws

    def train_ws_release(self, ws: th.Tensor) -> None:
        key = getattr(ws, "_marl_key", None)
        if key == (self._cfg_key, _tune_epoch) and len(self._ws_pool.setdefault(key, [])) < 2:
            self._ws_pool[key].append(ws)

    def _table(self, tensors: Dict[str, th.Tensor]) -> "C.Array":
        arr = (C.c_void_p * MARL_NPARAMS)()
        for name, slot in self.slots.items():
            t = tensors[name]
            if not t.is_cuda or t.dtype != th.float32 or not t.is_contiguous():
                raise RuntimeError(f"parameter {name} must be a contiguous fp32 GPU tensor")
            arr[slot] = t.data_ptr()
        return arr

    def set_heads_event(self, hip_event: Optional[int]) -> None:
        """marl_backward_heads_event: the event episode_backward records once the heads' parameter gradients are
        final (parallel.BucketedGradAllReduce); None clears it."""
        check(self.lib.marl_backward_heads_event(hip_event))

    # -- calls ----------------------------------------------------------------------
    def pack(self, params: Dict[str, th.Tensor]) -> None:
        """marl_pack_weights: refresh the padded / transposed copies after an update."""
        assert self.cfg is not None, "configure() first"
        wws = self.weights_ws()
        check(self.lib.marl_pack_weights(C.byref(self.cfg), self._table(params), wws.data_ptr(), _nbytes(wws),
                                         _stream(self.device)))
        self.pack_generation += 1
        self._packed_generation = self.weights_generation
        self._packed_params = params

    def episode_forward(
        self, img: th.Tensor, pos0: th.Tensor, h0: th.Tensor, c0: th.Tensor, hc0: th.Tensor,
        cc0: th.Tensor, noise: Optional[th.Tensor], forced_actions: Optional[th.Tensor] = None,
        train: bool = True, rng: Optional[Tuple[int, int]] = None,
        out: Optional[EpisodeTensors] = None, counters: Optional[th.Tensor] = None,
        ws: Optional[th.Tensor] = None,
    ) -> EpisodeTensors:
        """``noise`` = injected Exp(1) draws [Ns,Na,Nb,nA] (parity mode); ``noise=None`` with
        ``rng=(seed, offset)`` draws them inside the sampling kernel (perf mode)."""
        cfg = self.cfg
        assert cfg is not None
        na, nb, ns = cfg.nb_agents, cfg.batch, cfg.nb_steps
        dev = self.device
        img = _need(img, th.uint8 if cfg.img_u8 else th.float32, "img")
        pos0 = _need(pos0, th.int64, "pos0")
        h0, c0, hc0, cc0 = (_need(t, th.float32, n) for t, n in
                            ((h0, "h0"), (c0, "c0"), (hc0, "hc0"), (cc0, "cc0")))
        if noise is not None:
            noise = _need(noise, th.float32, "noise")
        if forced_actions is not None:
            forced_actions = _need(forced_actions, th.int64, "forced_actions")
        if noise is None and forced_actions is None and rng is None:
            raise ValueError("episode_forward needs noise, forced_actions or rng=(seed, offset)")
        seed, offset = rng if rng is not None else (0, 0)
        if out is None:  # (graph capture passes persistent output tensors: nothing may allocate)
            out = self.new_outputs()
        wws, ews = self.packed_weights_ws(), (ws if ws is not None else self.episode_ws(train))
        check(self.lib.marl_episode_forward(
            C.byref(cfg), wws.data_ptr(), _nbytes(wws), ews.data_ptr(), _nbytes(ews),
            img.data_ptr(), pos0.data_ptr(), h0.data_ptr(), c0.data_ptr(), hc0.data_ptr(),
            cc0.data_ptr(), _ptr(noise), _ptr(forced_actions), seed & _U64, offset & _U64,
            _ptr(counters), out.step_preds.data_ptr(), out.step_log_probas.data_ptr(), out.step_values.data_ptr(),
            out.step_pos.data_ptr(), out.step_actions.data_ptr(), int(train), _stream(dev)))
        if train and ws is None:  # (an episode with its own workspace does not touch the engine's)
            self.fwd_generation += 1
            self._fwd_img = img
            self._fwd_key = self._cfg_key
        return out

    def new_outputs(self) -> EpisodeTensors:
        cfg = self.cfg
        assert cfg is not None
        na, nb, ns, dev = cfg.nb_agents, cfg.batch, cfg.nb_steps, self.device
        return EpisodeTensors(
            th.empty(ns, na, nb, cfg.nb_class, device=dev),
            th.empty(ns, na, nb, device=dev),
            th.empty(ns, na, nb, device=dev),
            th.empty(ns, na, nb, 2, dtype=th.int64, device=dev),
            th.empty(ns, na, nb, dtype=th.int64, device=dev),
        )

    def episode_backward(
        self, g_preds: Optional[th.Tensor], g_logp: Optional[th.Tensor],
        g_values: Optional[th.Tensor], grads: Dict[str, th.Tensor],
        generation: Optional[int] = None,
        ws: Optional[th.Tensor] = None, img: Optional[th.Tensor] = None,
    ) -> None:
        """Backward of the LAST training rollout.  `generation` (the value of
        ``fwd_generation`` right after that rollout) makes a stale call fail loudly: the saved
        activations live in the single training workspace, which a later rollout overwrites."""
        cfg = self.cfg
        assert cfg is not None
        if ws is not None:  # an episode that owns its workspace (autograd path)
            if getattr(ws, "_marl_key", None) != (self._cfg_key, _tune_epoch) or img is None:
                raise RuntimeError(
                    "episode_backward: the engine was re-configured (other batch size / image shape / "
                    "layout knob) since this episode's rollout - its workspace no longer fits")
            gp = None if g_preds is None else _need(g_preds, th.float32, "g_preds")
            gl = None if g_logp is None else _need(g_logp, th.float32, "g_logp")
            gv = None if g_values is None else _need(g_values, th.float32, "g_values")
            wws = self.packed_weights_ws()
            check(self.lib.marl_episode_backward(
                C.byref(cfg), wws.data_ptr(), _nbytes(wws), ws.data_ptr(), _nbytes(ws), img.data_ptr(), _ptr(gp),
                _ptr(gl), _ptr(gv), self._table(grads), _stream(self.device)))
            return
        if self._fwd_img is None or self._fwd_key != self._cfg_key:
            raise RuntimeError(
                "episode_backward without a matching training rollout: the engine was "
                "re-configured (other batch size / image shape) or never ran episode_forward("
                "train=True) - its workspace no longer holds this episode's activations")
        if generation is not None and generation != self.fwd_generation:
            raise RuntimeError(
                "episode_backward for an episode whose saved activations were overwritten by a "
                "later training rollout (one live episode per engine: call backward before the "
                "next run_episode, or accumulate gradients across backward calls instead)")
        gp = None if g_preds is None else _need(g_preds, th.float32, "g_preds")
        gl = None if g_logp is None else _need(g_logp, th.float32, "g_logp")
        gv = None if g_values is None else _need(g_values, th.float32, "g_values")
        wws, ews = self.packed_weights_ws(), self.episode_ws(True)
        check(self.lib.marl_episode_backward(
            C.byref(cfg), wws.data_ptr(), _nbytes(wws), ews.data_ptr(), _nbytes(ews),
            self._fwd_img.data_ptr(), _ptr(gp), _ptr(gl), _ptr(gv), self._table(grads),
            _stream(self.device)))

    def a2c_loss(
        self, out: EpisodeTensors, y: th.Tensor, gamma: float, phase: int = 0,
        bufs: Optional[Tuple[th.Tensor, ...]] = None,
    ) -> Tuple[th.Tensor, th.Tensor, th.Tensor, th.Tensor, th.Tensor]:
        """Returns (g_preds, g_logp, g_values, scalars[4], adv_stats[3] float64)."""
        cfg = self.cfg
        assert cfg is not None
        dev = self.device
        y = _need(y, th.int64, "y")
        if bufs is None:
            bufs = (
                th.empty_like(out.step_preds), th.empty_like(out.step_log_probas),
                th.empty_like(out.step_values), th.zeros(4, device=dev),
                th.zeros(3, dtype=th.float64, device=dev),
            )
        gp, gl, gv, sc, st = bufs
        ews = self.episode_ws(True)
        check(self.lib.marl_a2c_loss_fwd_bwd(
            C.byref(cfg), ews.data_ptr(), _nbytes(ews), out.step_preds.data_ptr(),
            out.step_log_probas.data_ptr(), out.step_values.data_ptr(), y.data_ptr(),
            C.c_float(gamma), gp.data_ptr(), gl.data_ptr(), gv.data_ptr(), sc.data_ptr(),
            st.data_ptr(), phase, _stream(dev)))
        return bufs

    def adam(
        self, params: th.Tensor, grads: th.Tensor, exp_avg: th.Tensor, exp_avg_sq: th.Tensor,
        step: int, lr: float, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
        grad_scale: float = 1.0, counters: Optional[th.Tensor] = None,
    ) -> None:
        for t in (params, grads, exp_avg, exp_avg_sq):
            if not t.is_cuda or t.dtype != th.float32 or not t.is_contiguous():
                raise RuntimeError("adam buffers must be contiguous fp32 GPU tensors")
        check(self.lib.marl_adam_step(
            params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
            params.numel(), step, lr, betas[0], betas[1], eps, grad_scale, _ptr(counters),
            _stream(self.device)))

    def step_forward(
        self, obs: th.Tensor, msg: th.Tensor, norm_pos: th.Tensor, h: th.Tensor, c: th.Tensor,
        hc: th.Tensor, cc: th.Tensor, noise: Optional[th.Tensor] = None,
        rng: Optional[Tuple[int, int]] = None,
    ) -> Tuple[th.Tensor, ...]:
        """marl_step_forward: (probs, values, preds, new_msg, h, c, hc, cc) in [Na,Nb,..];
        with ``noise`` ([Na,Nb,nA] ~ Exp(1), parity mode) or ``rng=(seed, offset)`` (in-kernel
        draws) also (actions int64, log-probs)."""
        cfg = self.cfg
        assert cfg is not None
        na, nb = cfg.nb_agents, cfg.batch
        dev = self.device
        ins = [_need(t, th.float32, n) for t, n in (
            (obs, "obs"), (msg, "msg"), (norm_pos, "norm_pos"), (h, "h"), (c, "c"), (hc, "hc"),
            (cc, "cc"))]
        outs = (
            th.empty(na, nb, cfg.nb_action, device=dev), th.empty(na, nb, device=dev),
            th.empty(na, nb, cfg.nb_class, device=dev), th.empty(na, nb, cfg.n_m, device=dev),
            th.empty(na, nb, cfg.n_b, device=dev), th.empty(na, nb, cfg.n_b, device=dev),
            th.empty(na, nb, cfg.n_a, device=dev), th.empty(na, nb, cfg.n_a, device=dev),
        )
        extra: Tuple[th.Tensor, ...] = ()
        nz = act = lp = None
        if noise is not None or rng is not None:
            nz = None if noise is None else _need(noise, th.float32, "noise")
            act = th.empty(na, nb, dtype=th.int64, device=dev)
            lp = th.empty(na, nb, device=dev)
            extra = (act, lp)
        seed, offset = rng if rng is not None else (0, 0)
        wws, ews = self.packed_weights_ws(), self.episode_ws(False)
        check(self.lib.marl_step_forward(
            C.byref(cfg), wws.data_ptr(), _nbytes(wws), ews.data_ptr(), _nbytes(ews),
            *[t.data_ptr() for t in ins], *[t.data_ptr() for t in outs], _ptr(nz), seed & _U64,
            offset & _U64, _ptr(act), _ptr(lp), _stream(dev)))
        return outs + extra

    def draw_episode(self, seed: int, offset: int, with_noise: bool = False,
                     into: Optional[Tuple[th.Tensor, ...]] = None,
                     counters: Optional[th.Tensor] = None):
        """marl_draw_episode: the reference's reset draws (positions, h, c, h^, c^; optionally the
        per-step Exp(1) noise) from the library's counter-based generator in ONE launch.
        Returns (pos0, h0, c0, hc0, cc0, noise-or-None); ``into`` = persistent tensors to fill."""
        cfg = self.cfg
        assert cfg is not None
        na, nb, ns, dev = cfg.nb_agents, cfg.batch, cfg.nb_steps, self.device
        if into is None:
            into = (
                th.empty(na, nb, 2, dtype=th.int64, device=dev),
                th.empty(na, nb, cfg.n_b, device=dev), th.empty(na, nb, cfg.n_b, device=dev),
                th.empty(na, nb, cfg.n_a, device=dev), th.empty(na, nb, cfg.n_a, device=dev),
                th.empty(ns, na, nb, cfg.nb_action, device=dev) if with_noise else None,
            )
        pos0, h0, c0, hc0, cc0, noise = into
        check(self.lib.marl_draw_episode(
            C.byref(cfg), seed & _U64, offset & _U64, _ptr(counters), pos0.data_ptr(),
            h0.data_ptr(), c0.data_ptr(), hc0.data_ptr(), cc0.data_ptr(), _ptr(noise), _stream(dev)))
        return into

    # -- device-side iteration counters + hipGraph capture (marl_counters_*, marl_graph_*) ------
    def new_counters(self) -> th.Tensor:
        return th.zeros(_lib.MARL_COUNTERS_BYTES // 8, dtype=th.int64, device=self.device)

    def counters_set(self, counters: th.Tensor, rng_offset: int, step: int, lr: float,
                     betas: Tuple[float, float] = (0.9, 0.999)) -> None:
        check(self.lib.marl_counters_set(counters.data_ptr(), rng_offset & _U64, step, lr, betas[0],
                                         betas[1], _stream(self.device)))

    def counters_tick(self, counters: th.Tensor, lr: float,
                      betas: Tuple[float, float] = (0.9, 0.999)) -> None:
        check(self.lib.marl_counters_tick(counters.data_ptr(), lr, betas[0], betas[1],
                                          _stream(self.device)))

    def graph_begin(self) -> None:
        check(self.lib.marl_graph_begin(_stream(self.device)))

    def graph_end(self) -> int:
        handle = C.c_void_p(0)
        check(self.lib.marl_graph_end(_stream(self.device), C.byref(handle)))
        return handle.value

    def graph_launch(self, handle: int) -> None:
        check(self.lib.marl_graph_launch(handle, _stream(self.device)))

    def debug_buffer(self, name: str, t: int, train: bool = True) -> th.Tensor:
        """View [R, ld] of a named per-step activation inside the episode workspace (tests)."""
        off, ld = C.c_int64(0), C.c_int(0)
        check(self.lib.marl_debug_buffer(C.byref(self.cfg), int(train), name.encode(), t,
                                         C.byref(off), C.byref(ld)))
        rows = self.cfg.nb_agents * self.cfg.batch
        ws = self.episode_ws(train)
        return ws[off.value: off.value + rows * ld.value].view(rows, ld.value)

    # -- standalone environment kernels ------------------------------------------------
    def patch_gather(self, img: th.Tensor, pos: th.Tensor, f: int) -> th.Tensor:
        return patch_gather(img, pos, f)

    def transition(self, pos: th.Tensor, actions: th.Tensor, table: Sequence[Sequence[int]],
                   sizes: Sequence[int], f: int) -> th.Tensor:
        return transition(pos, actions, table, sizes, f)


# Environment kernels need no model: module-level wrappers (core/environment.py uses them)
def patch_gather(img: th.Tensor, pos: th.Tensor, f: int) -> th.Tensor:
    """marl_patch_gather: obs[a,b] = img[b, :, p0:p0+f, p1:p1+f] (environment.py:95-126)."""
    lib = _lib.load()
    img = _need(img, th.float32, "img")
    pos = _need(pos, th.int64, "pos")
    na, nb, _ = pos.shape
    _, c, h, w = img.shape
    obs = th.empty(na, nb, c, f, f, device=img.device)
    check(lib.marl_patch_gather(img.data_ptr(), pos.data_ptr(), obs.data_ptr(), na, nb, c, h, w, f,
                                _stream(img.device)))
    return obs


def transition(pos: th.Tensor, actions: th.Tensor, table: Sequence[Sequence[int]],
               sizes: Sequence[int], f: int) -> th.Tensor:
    """marl_transition: bounded move (environment.py:56-66,128-150)."""
    lib = _lib.load()
    pos = _need(pos, th.int64, "pos")
    actions = _need(actions, th.int64, "actions")
    out = th.empty_like(pos)
    flat = (C.c_int32 * (2 * len(table)))(*[v for mv in table for v in mv])
    rows = pos.shape[0] * pos.shape[1]
    check(lib.marl_transition(pos.data_ptr(), actions.data_ptr(), out.data_ptr(), flat, len(table),
                              rows, sizes[0], sizes[1], f, _stream(pos.device)))
    return out


def normalize_positions(pos: th.Tensor, sizes: Sequence[int]) -> th.Tensor:
    """marl_normalize_positions: pos / size per dimension (environment.py:74-81)."""
    lib = _lib.load()
    pos = _need(pos, th.int64, "pos")
    out = th.empty(pos.shape, dtype=th.float32, device=pos.device)
    rows = pos.shape[0] * pos.shape[1]
    check(lib.marl_normalize_positions(pos.data_ptr(), out.data_ptr(), rows, sizes[0], sizes[1],
                                       _stream(pos.device)))
    return out
