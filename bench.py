"""Headline benchmark: agent-env-steps/sec of one full A2C training iteration (rollout +
loss + backward through all steps + Adam [+ RCCL gradient all-reduce]) on the RESISC45
16-agent / 16-step configuration (BASELINE.json; SURVEY section 8d "C3"), synthetic data.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; every rank
works on its own batch shard (weak scaling), gradients are all-reduced over RCCL.
Prints ONE JSON line on rank 0.

The model is the product's own ``ModelsWrapper`` (reference init recipe), every random draw of
the episode comes from the library's counter-based generator (no torch kernel in the timed
loop), and the line is only printed after an untimed post-check of the last iteration
(finite loss, positions inside the image, actions in range).
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch as th  # noqa: E402

# RESISC45 config of README.md:41 (column C3) - the headline workload
C3 = dict(ft_extr="resisc45", window=12, n_b=256, n_a=256, n_m=64, n_m_o=96, n_d=16,
          nb_class=45, nlb=384, nla=384)
NA, NS, IMG = 16, 16, (3, 256, 256)
GAMMA, LR = 0.99, 1e-4
# the other BASELINE.json configurations (--config; reported in DESIGN.md, not the bench line)
OTHER = {
    "c2": (dict(ft_extr="mnist", window=6, n_b=64, n_a=64, n_m=16, n_m_o=24, n_d=8, nb_class=10,
                nlb=96, nla=96), 3, 5, (3, 28, 28), 1024, "MNIST 28x28, 3 agents, 5 steps, f=6"),
    "c4": (dict(ft_extr="aid", window=24, n_b=256, n_a=256, n_m=64, n_m_o=96, n_d=16, nb_class=30,
                nlb=320, nla=320, actions=[[3, 0], [-3, 0], [0, 3], [0, -3]]), 16, 16,
           (3, 600, 600), 32, "AID 600x600, 16 agents, 16 steps, f=24"),
    "c5": (dict(ft_extr="aid", window=32, n_b=256, n_a=256, n_m=64, n_m_o=96, n_d=16, nb_class=45,
                nlb=384, nla=384, actions=[[4, 0], [-4, 0], [0, 4], [0, -4]]), 64, 32,
           (3, 1024, 1024), 32, "synthetic 1024x1024, 64 agents, 32 steps, f=32"),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)": v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
# fp32 products on the bf16 pipe (gemm_split.hip): six bf16 MFMA products per fp32 product
PEAK_F32_VIA_BF16X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md "HBM3E peak BW" (spec)

# kernel classes of marl_profile_begin (include/marl_hip.h)
CLASSES = {
    0: "gemm_nt[_split]_kernel<LSTM> (belief+action LSTM cells, fused cell epilogue)",
    1: "gemm_nt[_split]_kernel (activations x weights: batched heads, dX / dU products, in-loop W_hh)",
    2: "gemm_tn[_split]_kernel (weight gradients: contraction over all Ns*R rows)",
    3: "cnn_fwd_kernel (gather + conv/GroupNorm/SiLU stack, one launch per step)",
    4: "panel_fwd/bwd_kernel (message encoder / decoder, policy hidden layer, per step)",
    5: "cnn_dgrad + cnn_wgrad kernels (CNN backward, batched over all steps)",
}


def csrc_sha256() -> str:
    """Fingerprint of the kernel sources (csrc/*.hip, *.h): ties PMC-derived numbers to a build."""
    import glob
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "marlclassification_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def cnn_shapes(cfg: dict):
    """(cin, cout, hin, hout) per conv layer of the feature extractor (networks/vision.py:23-52)."""
    from marlclassification_amd.engine import CNN_SPECS

    ch, _ = CNN_SPECS[cfg["ft_extr"]]
    out, h = [], cfg["window"]
    for ci, co in zip(ch[:-1], ch[1:]):
        ho = (h - 1) // 2 + 1
        out.append((ci, co, h, ho))
        h = ho
    return out


def algorithmic_work(cfg: dict, c_img: int):
    """Forward FLOPs (2 * MACs of every conv / linear) and compulsory HBM bytes per
    agent-env-step, as SURVEY section 8(d) defines them, plus the same split by kernel class."""
    conv = cnn_shapes(cfg)
    f_cnn = sum(2 * ho * ho * co * 9 * ci for ci, co, _, ho in conv)
    nf = conv[-1][1] * conv[-1][3] ** 2
    n_b, n_a, n_m, n_mo, n_d = cfg["n_b"], cfg["n_a"], cfg["n_m"], cfg["n_m_o"], cfg["n_d"]
    nlb, nla, nC = cfg["nlb"], cfg["nla"], cfg["nb_class"]
    nA = len(cfg.get("actions", [[1, 0], [-1, 0], [0, 1], [0, -1]]))
    nin = nf + n_mo + n_d
    f_lstm = 2 * (4 * n_b * (nin + n_b) + 4 * n_a * (nin + n_a))
    f_msg = 2 * (n_m * 2 * n_m + 2 * n_m * n_mo) + 2 * (n_b * 2 * n_m + 2 * n_m * n_m)  # decode + encode
    f_pol_hidden = 2 * n_a * nla
    f_heads = 2 * (n_a * nla + nla) + 2 * (n_b * nlb + nlb * nC) + 2 * nla * nA  # critic, predict, policy out
    f_pos = 2 * 2 * n_d
    fwd = f_cnn + f_lstm + f_msg + f_pol_hidden + f_heads + f_pos
    c_used = conv[0][0]
    patch = 4 * c_used * cfg["window"] ** 2
    by_fwd = patch + 2 * 4 * (2 * n_b + 2 * n_a) + 8 * n_m + 40 + 4 * nA + 4 * (nC + 2) + 16
    # activations kept for backward, written once + read once (fp32): conv pre-norm outputs and
    # statistics, U, message nets, gates, head hidden layers
    saved = (sum(ho * ho * co for _, co, _, ho in conv) + nin + 3 * 2 * n_m + n_m + n_mo + n_d + 4 * n_b
             + 4 * n_a + 2 * (2 * n_m) + n_m + 3 * nla + 2 * nlb + nA)
    by_train = by_fwd + 2 * 4 * saved
    # per kernel class, per agent-env-step, full training iteration (backward = dX + dW products)
    cls = {
        0: f_lstm,
        # NT: batched heads forward + every dX product of backward (the LSTM's dU / dh included)
        1: (2 * (n_a * nla) + 2 * (n_b * nlb + nlb * nC)) + (2 * nC * nlb + 2 * nlb * n_b + 2 * 2 * nla * n_a)
           + 2 * (4 * n_b * n_b + 4 * n_a * n_a) + 2 * (4 * n_b + 4 * n_a) * (nin),
        # TN: every weight gradient outside the CNN
        2: f_lstm + (2 * 2 * n_a * nla + 2 * nla * nA + 2 * nla) + 2 * (n_b * nlb + nlb * nC) + f_msg + f_pos,
        3: f_cnn,
        4: 3 * (f_msg + f_pol_hidden) - f_pol_hidden,  # forward + backward dX (weight grads are class 2)
        5: 2 * f_cnn,
    }
    # the same for a rollout (no-grad episode): only the forward products of each class run
    cls_fwd = {0: f_lstm, 1: 2 * (n_a * nla) + 2 * (n_b * nlb + nlb * nC), 3: f_cnn, 4: f_msg + f_pol_hidden}
    return fwd, by_fwd, by_train, cls, cls_fwd


def cpu_baseline(budget_s: float = 25.0) -> dict:
    """The oracle ("port" of the reference's CPU path, incl. its mask + masked_select crop)
    timed on this host's cores on a reduced batch of the same workload (CPU steps/s is
    batch-independent: SURVEY section 6).  Test infrastructure used as the measured CPU leg only."""
    from oracle import marl_oracle as mo

    nb = 8
    cfg = mo.OracleConfig(C3["ft_extr"], C3["window"], C3["n_b"], C3["n_a"], C3["n_m"],
                          C3["n_m_o"], C3["n_d"], C3["nb_class"], C3["nlb"], C3["nla"])
    params = mo.init_params(cfg, 0)
    img = th.rand(nb, *IMG, generator=th.Generator().manual_seed(0))
    y = th.randint(0, C3["nb_class"], (nb,), generator=th.Generator().manual_seed(1))
    m = {k: th.zeros_like(v) for k, v in params.items()}
    v = {k: th.zeros_like(x) for k, x in params.items()}

    def run(faithful: bool, budget: float, max_it: int) -> tuple:
        """mean of the timed iterations: the first is discarded, then AT LEAST three are timed (round 3 timed as
        few as one at batch 4 and the figure moved +-12 % run to run), more while the budget lasts"""
        times = []
        t_start = time.perf_counter()
        it = 0
        while True:
            inp = mo.draw_episode_inputs(cfg, NA, nb, NS, IMG[1:], 42 + it)
            t0 = time.perf_counter()
            _, _, grads = mo.train_iteration(params, cfg, img, y, inp, NS, GAMMA, faithful_crop=faithful)
            mo.adam_step(params, grads, m, v, it + 1, LR)
            times.append(time.perf_counter() - t0)
            it += 1
            if (it >= 4 and time.perf_counter() - t_start > budget) or it >= max_it:
                break
        return sum(times[1:]) / len(times[1:]), len(times)

    best, n_it = run(True, budget_s, 8)
    best_gather, n_it_g = run(False, 10.0, 8)  # same path with an O(f^2) index-gather crop
    return {
        "value": nb * NA * NS / best,
        "value_gather_crop": nb * NA * NS / best_gather,
        "unit": "agent-env-steps/s",
        "cores": th.get_num_threads(),
        "kind": "port",
        "sample": f"{n_it} full train iterations (1st discarded) of the same config at batch {nb}, "
                  "reference-faithful mask+masked_select crop, torch-CPU fp32; value_gather_crop = "
                  f"the same with an O(f^2) index-gather crop ({n_it_g} iterations, 1st discarded)",
    }


def matrix_products_note(lib, cfg) -> str:
    """what the matrix kernels of THIS run do: the library's own decisions for this configuration
    (marl_plan_query), not a re-derivation of them"""
    if lib.marl_tune_get(b"mfma_split", 1) == 0:
        return "exact fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32, gemm.hip): knob mfma_split = 0"

    def plan(key: bytes) -> bool:
        v = C.c_int(0)
        if lib.marl_plan_query(C.byref(cfg), 1, key, C.byref(v)) != 0:
            raise SystemExit("marl_plan_query failed: " + lib.marl_last_error().decode())
        return bool(v.value)

    note = ("fp32 results from three-term bf16 splits: six bf16 MFMA products per fp32 product, fp32 accumulate "
            "(gemm_split.hip: fp32 operands split while staged)")
    if plan(b"g3"):
        on = ["in-loop backward batch, batched heads, dU"]
        if plan(b"g3_lstm"):
            on.insert(0, "LSTM cells")
        if plan(b"g3_tn"):
            on.append("the four large weight gradients" + (" (one launch per cell" + (", phase-pipelined step" if plan(b"g3_tn_pipe") else "")
                                                            + ")" if plan(b"g3_tn_cell") else ""))
        note += ("; " + ", ".join(on) + " on operand images pre-split by their producers and staged by LDS-DMA "
                 "(gemm3.hip; marl_plan_query)")
        if plan(b"small_r"):
            note += "; small-batch tile plans (tiles < 2 x CUs)"
        if plan(b"wgrad3"):
            note += "; conv weight gradients with >= 16 input channels on the same six-product arithmetic (cnn_wgrad3_kernel)"
    return note


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None,
                    help="images per GPU (weak scaling); default: 256 for c3, the config's own otherwise (c4 / c5: 32; "
                         "BASELINE configs[3]'s weak-scaling variant is --config c4 --batch 256)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: this many images in all, N / world per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rollout-only", action="store_true")
    ap.add_argument("--torch-rng", action="store_true",
                    help="draw positions / states / noise with torch kernels (round-1 behaviour)")
    ap.add_argument("--graph", action="store_true", help="replay the iteration as a captured hipGraph")
    ap.add_argument("--config", choices=["c3"] + sorted(OTHER), default="c3")
    args = ap.parse_args()
    if args.graph and args.rollout_only:
        ap.error("--graph replays the whole training iteration: not with --rollout-only")
    global C3, NA, NS, IMG
    workload = "RESISC45 256x256x3, 16 agents, 16 steps, f=12, README dims (configs[2])"
    if args.config != "c3":
        C3, NA, NS, IMG, default_batch, workload = OTHER[args.config]
        if args.batch is None:
            args.batch = default_batch
        args.no_cpu_baseline = True
    if args.batch is None:
        args.batch = 256

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N "
                         "--master-addr 127.0.0.1 bench.py --gpus N ...")
    th.cuda.set_device(local_rank)
    dev = th.device("cuda", local_rank)

    import torch.distributed as dist

    distributed = "RANK" in os.environ  # launched by torch.distributed.run (any world size)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    from marlclassification_amd import _lib
    from marlclassification_amd.fused import FusedA2C, draw_episode, draw_episode_device
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import CNN_BY_NAME
    from marlclassification_amd.parallel import BucketedGradAllReduce, broadcast_parameters, shard_seed

    lib = _lib.load()
    actions = C3.get("actions", [[1, 0], [-1, 0], [0, 1], [0, -1]])
    th.manual_seed(0)  # reference init recipe (networks/init.py), same weights on every rank
    model = ModelsWrapper(CNN_BY_NAME[C3["ft_extr"]](C3["window"]), C3["n_b"], C3["n_a"], C3["n_m"],
                          C3["n_m_o"], C3["n_d"], 2, len(actions), C3["nb_class"], C3["nlb"],
                          C3["nla"]).to(dev)
    flat = model.flat_state()
    if distributed:
        broadcast_parameters(flat.params)
    eng = model.hip_engine(actions)
    spec = eng.spec
    is_c3 = args.config == "c3"
    nb = args.batch
    if args.global_batch:
        if args.global_batch % world != 0:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by the world size {world}")
        nb = args.global_batch // world
    eng.configure(NA, nb, NS, IMG)
    hook = BucketedGradAllReduce(world, None, flat.offsets, flat.numel, dev) if distributed else None
    fa = FusedA2C(eng, flat, LR, GAMMA, allreduce=hook, use_graph=args.graph)

    gen = th.Generator(device=dev).manual_seed(shard_seed(0, rank))
    img = th.rand(nb, *IMG, device=dev, generator=gen)  # pre-resident in HBM
    y = th.randint(0, C3["nb_class"], (nb,), device=dev, generator=gen)
    egen = th.Generator(device=dev).manual_seed(shard_seed(42, rank))
    rng_seed = shard_seed(42, rank)
    state = {"it": 0, "out": None, "scalars": None}

    def one_step():
        if args.torch_rng:
            draws = draw_episode(spec, NA, nb, NS, IMG[1:], dev, egen)
        else:
            draws = None if args.graph else draw_episode_device(eng, rng_seed, state["it"])
        if args.rollout_only:
            state["out"] = fa.rollout(img, draws, False)
        elif args.graph:
            state["out"], state["scalars"] = fa.iteration_graph(img, y, rng_seed, state["it"])
        else:
            state["out"], state["scalars"] = fa.iteration(img, y, draws)
        state["it"] += 1

    def fence():
        if distributed:
            dist.barrier()
        th.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        t = th.tensor([dt], dtype=th.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    # ---- untimed post-check: a broken kernel must not post a number ------------------------
    out = state["out"]
    pos = out.step_pos
    ok = bool(th.isfinite(out.step_preds).all()) and bool(th.isfinite(out.step_values).all())
    ok = ok and bool((out.step_log_probas <= 0).all())
    ok = ok and bool((pos >= 0).all()) and bool((pos[..., 0] + C3["window"] <= IMG[1]).all())
    ok = ok and bool((pos[..., 1] + C3["window"] <= IMG[2]).all())
    ok = ok and bool((out.step_actions >= 0).all()) and bool((out.step_actions < len(actions)).all())
    if state["scalars"] is not None:
        ok = ok and bool(th.isfinite(state["scalars"]).all())
        ok = ok and bool(th.isfinite(flat.params).all())
    if not ok:
        raise SystemExit("bench post-check failed: non-finite outputs or positions / actions out of range")

    # ---- rooflines ---------------------------------------------------------------------------
    # (1) whole iteration against both rooflines: algorithmic FLOPs / bytes of the
    #     timed work vs the fp32-MFMA and HBM peaks; (2) the dominant kernel class, found by a
    #     per-class HIP-event sweep (one extra iteration per class, events on the launch stream).
    # Every rank runs the extra iterations (they contain the gradient all-reduce).
    roofline = None
    cpu = None
    f_fwd, by_fwd, by_train, cls_train, cls_fwd = algorithmic_work(C3, IMG[0])
    cls_flops = cls_fwd if args.rollout_only else cls_train  # what each class computes in the timed work
    steps_per_iter_gpu = nb * NA * NS
    sweep = {}

    def sweep_step():
        # the per-class events are recorded by the launchers: a hipGraph replay runs none of
        # them, so the sweep iterations are launched eagerly whatever the timed loop did
        if not args.graph:
            return one_step()
        state["out"], state["scalars"] = fa.iteration(img, y, draw_episode_device(eng, rng_seed, 1 << 30))

    for c in CLASSES:
        if c not in cls_flops:  # (rollout: no weight-gradient / CNN-backward launches)
            continue
        if rank == 0:
            lib.marl_profile_begin(c, 4096)
        sweep_step()
        if rank == 0:
            tot, cnt = C.c_double(0), C.c_int(0)
            lib.marl_profile_end(C.byref(tot), C.byref(cnt))
            sweep[c] = (tot.value, cnt.value)
    if rank == 0:
        t_iter = dt / args.steps
        mult = 1.0 if args.rollout_only else 3.0
        flops_iter = mult * f_fwd * steps_per_iter_gpu
        bytes_iter = (by_fwd if args.rollout_only else by_train) * steps_per_iter_gpu
        t_mfma = flops_iter / (PEAK_F32_MFMA_TFLOPS * 1e12)
        t_hbm = bytes_iter / (PEAK_HBM_GBS * 1e9)
        dom = max(sweep, key=lambda c: sweep[c][0])
        dom_ms, dom_n = sweep[dom]
        dom_flops = cls_flops[dom] * steps_per_iter_gpu
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # the matrix classes 0-2 run fp32 products as six bf16 MFMA products (mfma_split, default):
        # their pipe is the bf16 one, priced in fp32-equivalent FLOPs = 2.5 PF / 6; with the knob
        # off (and for the CNN / panel classes, which use v_mfma_f32_*) the fp32-MFMA peak applies
        split_on = lib.marl_tune_get(b"mfma_split", 1) != 0
        peak = PEAK_F32_VIA_BF16X6_TFLOPS if (split_on and dom in (0, 1, 2)) else PEAK_F32_MFMA_TFLOPS
        roofline = {
            "kernel": CLASSES[dom],
            "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "peak_note": ("fp32-equivalent: fp32 operands split into three bf16 terms, six bf16 MFMA "
                          "products per fp32 product = 2500 / 6 TF; against the 157.3 TF of "
                          "v_mfma_f32_32x32x2_f32 this is frac %.3f" % (achieved / PEAK_F32_MFMA_TFLOPS))
                         if peak != PEAK_F32_MFMA_TFLOPS else "v_mfma_f32_* (exact fp32 MFMA)",
            "avg_launch_us": round(dom_ms * 1e3 / max(1, dom_n), 2), "launches": dom_n,
            "share_of_iteration": round(dom_ms * 1e-3 / t_iter, 4),
            "traffic": None,
            "classes": {CLASSES[c].split(" ")[0]: {"ms": round(ms, 3), "launches": n,
                                                   "tflops": round(cls_flops[c] * steps_per_iter_gpu
                                                                   / max(ms, 1e-9) / 1e9, 1)}
                        for c, (ms, n) in sweep.items()},
            "iteration": {
                "flops": flops_iter, "bytes": bytes_iter,
                "t_mfma_ms": round(t_mfma * 1e3, 3), "t_hbm_ms": round(t_hbm * 1e3, 3),
                "t_measured_ms": round(t_iter * 1e3, 3),
                "achieved": round(max(t_mfma, t_hbm) / t_iter, 4),
                "bound": "mfma" if t_mfma >= t_hbm else "hbm",
                "note": "max(t_hbm, t_mfma) / t_measured with algorithmic FLOPs (3x forward) and "
                        "bytes per agent-env-step from SURVEY 8(d); t_mfma is priced at the EXACT-fp32 "
                        "MFMA peak (157.3 TF) whatever pipe the products run on",
            },
        }
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of
        # THIS round (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), if they cover it
        # (only if they were taken from THIS library: the file carries the sha256 of csrc/*.hip|*.h)
        tpath = os.path.join(ROOT, "profiles", "r06_traffic.json")
        if nb == 256 and is_c3 and not args.rollout_only and os.path.exists(tpath):
            with open(tpath, "r", encoding="utf-8") as f:
                tj = json.load(f)
            ent = tj.get(str(dom))
            if ent and tj.get("src_sha256") == csrc_sha256():
                roofline["traffic"] = ent["traffic_bytes_per_launch"]
                roofline["traffic_unit"] = "bytes/launch (PMC, profiles/r06_traffic.json, sources match)"
            elif ent:
                roofline["traffic_unit"] = ("null: profiles/r06_traffic.json was measured on other kernel "
                                            "sources (sha256 differs)")
    if distributed:
        dist.barrier()
    if rank == 0 and not args.no_cpu_baseline and args.gpus == 1:
        cpu = cpu_baseline()

    if rank == 0:
        steps_per_iter = nb * NA * NS * world
        value = steps_per_iter * args.steps / dt
        line = {
            "metric": "agent-env-steps/sec" + (" (rollout only)" if args.rollout_only else ""),
            "value": round(value, 1), "unit": "agent-env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload + ("; rollout only: one no-grad episode (EpisodeSampler.run_episode under "
                                        "th.no_grad), no loss / backward / optimiser" if args.rollout_only else
                                        "; full train iteration: rollout + A2C loss + BPTT backward + Adam"
                                        + (" + RCCL grad all-reduce" if world > 1 else "")),
                "batch_per_gpu": nb, "global_batch": nb * world,
                "parallelism": f"dp{world}",
                "rng": "torch" if args.torch_rng else "library (Philox4x32-10)",
                "matrix_products": matrix_products_note(lib, eng.cfg),
                "launch": "hipGraph replay" if args.graph else "eager",
            },
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
