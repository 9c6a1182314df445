"""ctypes binding of libmarl_hip.so (the C ABI declared in include/marl_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it has not
been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C marlclassification_amd/csrc``) loading fails loudly.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

MARL_MAX_CNN_LAYERS = 5
MARL_MAX_ACTIONS = 16
MARL_ABI_VERSION = 3
MARL_COUNTERS_BYTES = 32

_HERE = os.path.dirname(os.path.abspath(__file__))
# MARL_LIB_PATH: another build of the SAME library (tools/build_variant.sh: timing / ablation builds, the previous
# round's library for same-box A/Bs) - never a different implementation: load() checks the ABI and every export
LIB_PATH = os.environ.get("MARL_LIB_PATH") or os.path.join(_HERE, "csrc", "libmarl_hip.so")


class MarlConfig(C.Structure):
    """Mirror of ``marl_config`` (include/marl_hip.h)."""

    _fields_ = [
        ("nb_agents", C.c_int32),
        ("batch", C.c_int32),
        ("nb_steps", C.c_int32),
        ("img_c", C.c_int32),
        ("img_h", C.c_int32),
        ("img_w", C.c_int32),
        ("window", C.c_int32),
        ("cnn_layers", C.c_int32),
        ("cnn_ch", C.c_int32 * (MARL_MAX_CNN_LAYERS + 1)),
        ("cnn_groups", C.c_int32 * MARL_MAX_CNN_LAYERS),
        ("n_b", C.c_int32),
        ("n_a", C.c_int32),
        ("n_m", C.c_int32),
        ("n_m_o", C.c_int32),
        ("n_d", C.c_int32),
        ("nb_action", C.c_int32),
        ("nb_class", C.c_int32),
        ("nlb", C.c_int32),
        ("nla", C.c_int32),
        ("actions", (C.c_int32 * 2) * MARL_MAX_ACTIONS),
        ("img_u8", C.c_int32),
    ]


# parameter table indices (enum in include/marl_hip.h)
_NAMES = (
    "POS_W POS_B POS_LNW POS_LNB "
    "ENC_W0 ENC_B0 ENC_LN0W ENC_LN0B ENC_W1 ENC_B1 ENC_LN1W ENC_LN1B "
    "DEC_W0 DEC_B0 DEC_LN0W DEC_LN0B DEC_W1 DEC_B1 DEC_LN1W DEC_LN1B "
    "LB_WIH LB_WHH LB_BIH LB_BHH LA_WIH LA_WHH LA_BIH LA_BHH "
    "POL_W0 POL_B0 POL_LNW POL_LNB POL_W1 POL_B1 "
    "CRI_W0 CRI_B0 CRI_LNW CRI_LNB CRI_W1 CRI_B1 "
    "PRE_W0 PRE_B0 PRE_LNW PRE_LNB PRE_W1 PRE_B1"
).split()
P = {name: 20 + i for i, name in enumerate(_NAMES)}
MARL_NPARAMS = 20 + len(_NAMES)

# every symbol include/marl_hip.h declares
EXPORTS = (
    "marl_abi_version marl_last_error marl_param_numel marl_workspace_sizes marl_pack_weights "
    "marl_patch_gather marl_transition marl_episode_forward marl_episode_backward "
    "marl_a2c_loss_fwd_bwd marl_adam_step marl_step_forward marl_gemm_nt marl_gemm_tn "
    "marl_gemm_tn_scratch marl_gemm_nt_weights marl_gemm_weight_image_bytes marl_ln_silu_fwd marl_debug_buffer "
    "marl_profile_begin marl_profile_end marl_normalize_positions "
    "marl_cnn_wgrad marl_cnn_wgrad_scratch marl_tune marl_tune_get marl_draw_episode "
    "marl_counters_set marl_counters_tick marl_graph_begin marl_graph_end marl_graph_launch "
    "marl_graph_destroy marl_image_bytes marl_image_build marl_gemm_nt_images marl_gemm_nt_images_batch "
    "marl_lstm_images marl_gemm_tn_images marl_gemm_tn_images_scratch marl_plan_query "
    "marl_gemm_tn_images_cell marl_gemm_tn_images_cell_scratch marl_backward_heads_event"
).split()

_lib: Optional[C.CDLL] = None

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_u64 = C.c_uint64
_cfgp = C.POINTER(MarlConfig)


def _declare(lib: C.CDLL) -> None:
    lib.marl_abi_version.restype = _i
    lib.marl_last_error.restype = C.c_char_p
    lib.marl_param_numel.restype = _i64
    lib.marl_param_numel.argtypes = [_cfgp, _i]
    lib.marl_workspace_sizes.argtypes = [_cfgp, _i, C.POINTER(_sz), C.POINTER(_sz)]
    lib.marl_pack_weights.argtypes = [_cfgp, C.POINTER(_vp), _vp, _sz, _vp]
    lib.marl_patch_gather.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.marl_transition.argtypes = [_vp, _vp, _vp, C.POINTER(C.c_int32), _i, _i, _i, _i, _i, _vp]
    lib.marl_episode_forward.argtypes = ([_cfgp, _vp, _sz, _vp, _sz] + [_vp] * 8 + [_u64, _u64, _vp] + [_vp] * 5 +
                                         [_i, _vp])
    lib.marl_draw_episode.argtypes = [_cfgp, _u64, _u64, _vp] + [_vp] * 7
    lib.marl_counters_set.argtypes = [_vp, _u64, _i64, _f, _f, _f, _vp]
    lib.marl_counters_tick.argtypes = [_vp, _f, _f, _f, _vp]
    lib.marl_graph_begin.argtypes = [_vp]
    lib.marl_graph_end.argtypes = [_vp, C.POINTER(_vp)]
    lib.marl_graph_launch.argtypes = [_vp, _vp]
    lib.marl_graph_destroy.argtypes = [_vp]
    lib.marl_episode_backward.argtypes = [_cfgp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, C.POINTER(_vp), _vp]
    lib.marl_a2c_loss_fwd_bwd.argtypes = (
        [_cfgp, _vp, _sz, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp]
    )
    lib.marl_adam_step.argtypes = [_vp, _vp, _vp, _vp, _i64, _i64, _f, _f, _f, _f, _f, _vp, _vp]
    lib.marl_step_forward.argtypes = ([_cfgp, _vp, _sz, _vp, _sz] + [_vp] * 15 + [_vp, _u64, _u64, _vp, _vp] +
                                      [_vp])
    lib.marl_normalize_positions.argtypes = [_vp, _vp, _i, _i, _i, _vp]
    lib.marl_gemm_nt.argtypes = [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.marl_gemm_nt_weights.argtypes = [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]
    lib.marl_gemm_weight_image_bytes.restype = _sz
    lib.marl_gemm_weight_image_bytes.argtypes = [_i, _i]
    lib.marl_gemm_tn.argtypes = [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i64, _vp, _sz, _vp]
    lib.marl_gemm_tn_scratch.restype = _sz
    lib.marl_gemm_tn_scratch.argtypes = [_i, _i, _i64]
    lib.marl_ln_silu_fwd.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]
    lib.marl_image_bytes.restype = _sz
    lib.marl_image_bytes.argtypes = [_i64, _i]
    lib.marl_image_build.argtypes = [_vp, _i, _i64, _i, _vp, _vp]
    lib.marl_gemm_nt_images.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.marl_gemm_nt_images_batch.argtypes = [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i),
                                              C.POINTER(_i), _i, _i, _i, _i, _vp]
    lib.marl_gemm_tn_images_scratch.restype = _sz
    lib.marl_gemm_tn_images_scratch.argtypes = [_i, _i, _i64]
    lib.marl_gemm_tn_images.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i64, _vp, _vp, _sz, _vp]
    lib.marl_lstm_images.argtypes = [_vp, _i] + [_vp] * 9 + [_i] * 6 + [_vp]
    lib.marl_cnn_wgrad.argtypes = ([_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64] + [_i] * 8 +
                                   [_vp, _vp, _vp, _sz, _vp])
    lib.marl_cnn_wgrad_scratch.restype = _sz
    lib.marl_cnn_wgrad_scratch.argtypes = [_i64, _i, _i, _i, _i, _i]
    lib.marl_tune.argtypes = [C.c_char_p, _i]
    lib.marl_tune_get.argtypes = [C.c_char_p, _i]
    lib.marl_profile_begin.argtypes = [_i, _i]
    lib.marl_profile_end.argtypes = [C.POINTER(C.c_double), C.POINTER(_i)]
    lib.marl_debug_buffer.argtypes = [_cfgp, _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i)]
    lib.marl_gemm_tn_images_cell_scratch.restype = _sz
    lib.marl_gemm_tn_images_cell_scratch.argtypes = [_i, _i, _i, _i64]
    lib.marl_gemm_tn_images_cell.argtypes = [_vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp]
    lib.marl_plan_query.argtypes = [_cfgp, _i, C.c_char_p, C.POINTER(_i)]
    lib.marl_backward_heads_event.argtypes = [_vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("marl_abi_version", "marl_tune_get"):
            fn.restype = _i


def load() -> C.CDLL:
    """Loads libmarl_hip.so; raises if it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library is the only implementation of this "
            "package (no CPU fallback). Build it with `make -C marlclassification_amd/csrc` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} lacks symbols {missing}")
    _declare(lib)
    if lib.marl_abi_version() != MARL_ABI_VERSION:
        raise RuntimeError("libmarl_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().marl_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libmarl_hip error {rc}: {msg}")
