"""Headline benchmark: agent-env-steps/sec of one full A2C training iteration (rollout +
loss + backward through all steps + Adam [+ RCCL gradient all-reduce]) on the RESISC45
16-agent / 16-step configuration (BASELINE.json; SURVEY section 8d "C3"), synthetic data.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; every rank
works on its own batch shard (weak scaling), gradients are all-reduced over RCCL.
Prints ONE JSON line on rank 0.
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
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0


def lstm_flops_per_launch(rows: int) -> float:
    """Algorithmic FLOPs of the fused belief+action LSTM launch:
    2 * rows * sum over both cells of 4n * (nin + n)."""
    nf = 64 * 2 * 2
    nin = nf + C3["n_m_o"] + C3["n_d"]
    per_row = 2 * (4 * C3["n_b"] * (nin + C3["n_b"]) + 4 * C3["n_a"] * (nin + C3["n_a"]))
    return float(per_row) * rows


def cpu_baseline(budget_s: float = 25.0) -> dict:
    """The oracle ("port" of the reference's CPU path, incl. its mask + masked_select crop)
    timed on this host's cores on a reduced batch of the same workload (CPU steps/s is
    batch-independent: SURVEY section 6)."""
    from oracle import marl_oracle as mo

    nb = 4
    cfg = mo.OracleConfig(C3["ft_extr"], C3["window"], C3["n_b"], C3["n_a"], C3["n_m"],
                          C3["n_m_o"], C3["n_d"], C3["nb_class"], C3["nlb"], C3["nla"])
    params = mo.init_params(cfg, 0)
    img = th.rand(nb, *IMG, generator=th.Generator().manual_seed(0))
    y = th.randint(0, C3["nb_class"], (nb,), generator=th.Generator().manual_seed(1))
    m = {k: th.zeros_like(v) for k, v in params.items()}
    v = {k: th.zeros_like(x) for k, x in params.items()}

    def run(faithful: bool, budget: float, max_it: int) -> tuple:
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
            if it >= 2 and time.perf_counter() - t_start > budget or it >= max_it:
                break
        return (sum(times[1:]) / len(times[1:]) if len(times) > 1 else times[0]), len(times)

    best, n_it = run(True, budget_s, 6)
    best_gather, _ = run(False, 8.0, 4)  # same path with an O(f^2) index-gather crop
    return {
        "value": nb * NA * NS / best,
        "value_gather_crop": nb * NA * NS / best_gather,
        "unit": "agent-env-steps/s",
        "cores": th.get_num_threads(),
        "kind": "port",
        "sample": f"{n_it} full train iterations (1st discarded) of the same config at batch {nb}, "
                  "reference-faithful mask+masked_select crop, torch-CPU fp32; value_gather_crop = "
                  "the same with an O(f^2) index-gather crop",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rollout-only", action="store_true")
    ap.add_argument("--config", choices=["c3"] + sorted(OTHER), default="c3")
    args = ap.parse_args()
    global C3, NA, NS, IMG
    workload = "RESISC45 256x256x3, 16 agents, 16 steps, f=12, README dims (configs[2])"
    if args.config != "c3":
        C3, NA, NS, IMG, default_batch, workload = OTHER[args.config]
        if args.batch == 256:
            args.batch = default_batch
        args.no_cpu_baseline = True

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
    from marlclassification_amd.engine import HipEngine, ModelSpec
    from marlclassification_amd.fused import FlatParams, FusedA2C, draw_episode
    from marlclassification_amd.parallel import GradAllReduce, shard_seed
    from oracle.marl_oracle import OracleConfig, init_params, param_shapes

    lib = _lib.load()
    spec = ModelSpec(**C3)
    is_c3 = args.config == "c3"
    eng = HipEngine(spec, dev)
    nb = args.batch
    eng.configure(NA, nb, NS, IMG)
    ocfg = OracleConfig(C3["ft_extr"], C3["window"], C3["n_b"], C3["n_a"], C3["n_m"], C3["n_m_o"],
                        C3["n_d"], C3["nb_class"], C3["nlb"], C3["nla"],
                        actions=C3.get("actions", [[1, 0], [-1, 0], [0, 1], [0, -1]]))
    flat = FlatParams(param_shapes(ocfg), dev)
    flat.load(init_params(ocfg, 0))  # reference init recipe (networks/init.py), same on all ranks
    hook = GradAllReduce(world) if distributed else None
    fa = FusedA2C(eng, flat, LR, GAMMA, allreduce=hook)

    gen = th.Generator(device=dev).manual_seed(shard_seed(0, rank))
    img = th.rand(nb, *IMG, device=dev, generator=gen)  # pre-resident in HBM
    y = th.randint(0, C3["nb_class"], (nb,), device=dev, generator=gen)
    egen = th.Generator(device=dev).manual_seed(shard_seed(42, rank))

    def one_step():
        draws = draw_episode(spec, NA, nb, NS, IMG[1:], dev, egen)
        if args.rollout_only:
            fa.rollout(img, draws, False)
        else:
            fa.iteration(img, y, draws)

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

    # ---- roofline of the dominant kernel (fused LSTM GEMM), HIP events on its stream ------
    # Every rank runs the extra iterations (they contain the gradient all-reduce); only rank 0
    # records events around its LSTM launches.
    roofline = None
    cpu = None
    nprof = 3
    if rank == 0:
        lib.marl_profile_begin(0, nprof * NS + 8)
    for _ in range(nprof):
        one_step()
    if rank == 0:
        tot, cnt = C.c_double(0), C.c_int(0)
        lib.marl_profile_end(C.byref(tot), C.byref(cnt))
        avg_s = tot.value / max(1, cnt.value) / 1e3
        achieved = lstm_flops_per_launch(NA * nb) / avg_s / 1e12 if is_c3 else 0.0
        roofline = {
            "kernel": "gemm_nt_kernel<128,128,4,1,LSTM> (belief+action LSTM cells, fused epilogue)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt.value, "traffic": None,
        }
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
        # correction + WRITE_SIZE; profiles/r01_lstm_traffic.json says how they were taken)
        tpath = os.path.join(ROOT, "profiles", "r01_lstm_traffic.json")
        if nb == 256 and is_c3 and os.path.exists(tpath):
            with open(tpath, "r", encoding="utf-8") as f:
                roofline["traffic"] = json.load(f)["traffic_bytes_per_launch"]
            roofline["traffic_unit"] = "bytes/launch (PMC, profiles/r01_lstm_traffic.json)"
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
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workload + "; full train iteration: rollout + A2C loss + BPTT "
                            "backward + Adam" + (" + RCCL grad all-reduce" if world > 1 else ""),
                "batch_per_gpu": nb, "global_batch": nb * world,
                "parallelism": f"dp{world}",
            },
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
