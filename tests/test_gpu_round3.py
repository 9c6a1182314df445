"""GPU, round 3: the exact multi-GPU command line (torch.distributed.run + RCCL at world size 1),
the full oracle gradient at the benched size, BASELINE.json's full sizes as property tests,
hipGraph replay mixed with eager iterations, and the input pipeline's decode rate.

Everything goes through the C ABI; subprocess tests start FRESH children (a process that has
touched the GPU must not exec another program on this pool)."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch as th

from oracle import marl_oracle as mo
from tests.util import Golden, model_spec

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATOL = 1e-5


def _child_env(port):
    env = dict(os.environ)
    env.update(PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    return env


# ---- (a) the driver's SCALE command line, at world size 1 ----------------------------------------
def test_bench_under_torch_distributed_run_uses_rccl(device):
    """`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` is what the
    driver launches for the 1/2/4/8-GPU curve.  N = 1 exercises every line of it on one GPU:
    "nccl" (= RCCL) process group, parameter broadcast, the all-reduce of the real flat gradient
    buffer inside the timed iteration, the rank-0 JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=_child_env(29517), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["config"]["parallelism"] == "dp1"
    assert "RCCL grad all-reduce" not in j["config"]["workload"]  # (the suffix names world > 1 only)
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["scaling"] == "weak"
    assert j["roofline"]["frac"] > 0 and j["roofline"]["iteration"]["t_measured_ms"] > 0
    # strong-scaling mode of the same command line
    r = subprocess.run(cmd[:-1] + ["--global-batch", "64", "--no-cpu-baseline"], cwd=ROOT,
                       env=_child_env(29518), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["scaling"] == "strong" and j["config"]["global_batch"] == 64 and j["config"]["batch_per_gpu"] == 64


def test_cli_train_under_torch_distributed_run(device, tmp_path):
    """The training driver under torch.distributed.run: RCCL process group, weight broadcast,
    index-sharded loader, all-reduce hook, rank-0 outputs."""
    out = tmp_path / "run"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29519", "-m", "marlclassification_amd"]
    cmd += (f"-a 3 --step 3 --cuda --run-id ddp train --ft-extr mnist --f 6 --img-size 28 --nb-class 10 "
            f"--nb 16 --na 16 --nm 8 --nmo 12 --nd 4 --nlb 24 --nla 24 --batch-size 16 --nb-epoch 1 "
            f"--lr 1e-3 --res-folder synthetic -o {out}").split()
    r = subprocess.run(cmd, cwd=ROOT, env=_child_env(29519), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "world size 1" in r.stdout and "epoch 0:" in r.stdout
    assert (out / "marl.json").exists() and (out / "models" / "nn_models_epoch_0.pt").exists()


# ---- (b) every entry of the gradient at the benched size -----------------------------------------
def test_full_gradient_at_benched_size_matches_oracle(device):
    """G4 (RESISC45 dims, 2 images) tiled 128x to B = 256 (R = 4096, NR = 65536: the plans
    bench.py runs).  With the advantage statistics of the un-tiled batch (loss phase 2) the tiled
    batch's gradient IS the fixture's; the oracle's full gradient of the 2-image batch (CPU,
    seconds) is compared entry by entry: all 1.69 M of them."""
    from marlclassification_amd.engine import HipEngine

    g = Golden("g4_resisc_b2")
    rep = 128
    _, lo, grads_ref = mo.train_iteration(g.params, g.cfg, g.img, g.y, g.inp, g.ns, g.gamma)
    i = g.inp
    eng1 = HipEngine(model_spec(g.cfg), device)
    eng1.configure(g.na, g.nb, g.ns, g.img.shape[1:])
    eng1.pack({k: v.to(device) for k, v in g.params.items()})
    small = [t.to(device) for t in (i.pos0, i.h0, i.c0, i.hc0, i.cc0, i.q)]
    out1 = eng1.episode_forward(g.img.to(device), *small, None, True)
    stats = eng1.a2c_loss(out1, g.y.to(device), g.gamma, phase=1)[4].clone()
    del eng1

    tile = lambda t, dim: th.cat([t] * rep, dim=dim)  # noqa: E731
    nb = g.nb * rep
    eng = HipEngine(model_spec(g.cfg), device)
    eng.configure(g.na, nb, g.ns, g.img.shape[1:])
    eng.pack({k: v.to(device) for k, v in g.params.items()})
    big = [tile(i.pos0, 1), tile(i.h0, 1), tile(i.c0, 1), tile(i.hc0, 1), tile(i.cc0, 1), tile(i.q, 2)]
    out = eng.episode_forward(tile(g.img, 0).to(device), *[t.to(device) for t in big], None, True)
    y = tile(g.y, 0).to(device)
    bufs = eng.a2c_loss(out, y, g.gamma, phase=1)
    bufs[4].copy_(stats)
    gp, gl, gv, sc, _ = eng.a2c_loss(out, y, g.gamma, phase=2, bufs=bufs)
    assert abs(sc[0].item() - lo.loss.item()) <= 5e-5 * max(1.0, abs(lo.loss.item()))
    grads = {k: th.zeros_like(v, device=device) for k, v in g.params.items()}
    eng.episode_backward(gp, gl, gv, grads)
    n, worst, bad = 0, 0.0, {}
    for k, ref in grads_ref.items():
        err = (grads[k].cpu().double() - ref.double()).abs().max().item()
        scale = ref.abs().max().item()
        n += ref.numel()
        worst = max(worst, err / scale if scale > 1e-12 else 0.0)
        if not err <= 1e-4 * scale + 1e-7:
            bad[k] = (err, scale)
    assert not bad, bad
    assert n > 1_600_000
    from tests.util import record

    record("full_gradient", {"entries": n, "max_err_over_tensor_scale": worst, "tolerance": 1e-4,
                             "margin": 1e-4 / worst if worst > 0 else None, "batch": nb})


# ---- (d) BASELINE.json's full sizes: properties the domain offers ----------------------------------
FULL = {
    # tag: (cfg dict of bench.py, agents, steps, image, batch)
    "c2_mnist_b1024": ("c2", 1024),
    "c4_aid_b32": ("c4", 32),
    "c5_synth_b32": ("c5", 32),
}


@pytest.mark.parametrize("tag", list(FULL))
def test_full_size_training_iterations_are_sane_and_replayable(device, tag):
    """configs[1] at B = 1024, configs[3] at 32 images per GPU, configs[4] at 32 per GPU: the row
    tile counts, split-K slab counts and (64 agents) two-launch panels these batch sizes select.
    Properties: finite outputs / loss / parameters, positions inside the image and consistent
    with the sampled moves, log-probs <= 0, probabilities of the sampled action consistent, the
    SAME (seed, offset) replays bit-identically, another offset does not."""
    import bench
    from marlclassification_amd.fused import FusedA2C, draw_episode_device
    from marlclassification_amd.networks import ModelsWrapper
    from marlclassification_amd.networks.vision import CNN_BY_NAME

    name, nb = FULL[tag]
    c, na, ns, shape, _, _ = bench.OTHER[name]
    actions = c.get("actions", [[1, 0], [-1, 0], [0, 1], [0, -1]])
    f = c["window"]

    def run(offsets):
        th.manual_seed(0)
        model = ModelsWrapper(CNN_BY_NAME[c["ft_extr"]](f), c["n_b"], c["n_a"], c["n_m"], c["n_m_o"],
                              c["n_d"], 2, len(actions), c["nb_class"], c["nlb"], c["nla"]).to(device)
        flat = model.flat_state()
        eng = model.hip_engine(actions)
        eng.configure(na, nb, ns, shape)
        fa = FusedA2C(eng, flat, 1e-4, 0.99)
        gen = th.Generator(device=device).manual_seed(5)
        img = th.rand(nb, *shape, device=device, generator=gen)
        y = th.randint(0, c["nb_class"], (nb,), device=device, generator=gen)
        res = []
        for off in offsets:
            out, sc = fa.iteration(img, y, draw_episode_device(eng, 9, off))
            res.append((out.step_pos.clone(), out.step_actions.clone(), out.step_preds.clone(),
                        out.step_log_probas.clone(), sc.clone()))
        th.cuda.synchronize()
        return res, flat.params.clone()

    (a0, a1), pa = run((0, 1))
    (b0, b1), pb = run((0, 1))
    pos, act, preds, logp, sc = a0
    assert bool(th.isfinite(preds).all()) and bool(th.isfinite(sc).all()) and bool(th.isfinite(pa).all())
    assert bool((logp <= 0).all()) and bool(th.isfinite(logp).all())
    assert bool((pos >= 0).all()) and bool((pos[..., 0] + f < shape[1] + 1).all())
    assert bool((pos[..., 1] + f < shape[2] + 1).all())
    assert bool((act >= 0).all()) and bool((act < len(actions)).all())
    # consecutive positions differ by the sampled move or not at all (a refused move)
    table = th.tensor(actions, device=device)
    step = pos[1:] - pos[:-1]
    mv = table[act[1:]]
    assert bool(((step == mv).all(-1) | (step == 0).all(-1)).all())
    # replay: same seeds -> same bits, whole run (two iterations incl. the Adam updates)
    for x, yv in zip(a0 + a1, b0 + b1):
        assert th.equal(x, yv)
    assert th.equal(pa, pb)
    assert not th.equal(a0[1], a1[1])  # another offset draws other actions


# ---- hipGraph replay interleaved with eager iterations / a changed learning rate -------
def test_graph_replay_survives_eager_iterations_and_lr_changes(device):
    from marlclassification_amd.fused import FlatParams, FusedA2C, draw_episode_device

    g = Golden("g2_mnist_c1")
    img, y = g.img.to(device), g.y.to(device)

    def run(mode):
        from marlclassification_amd.engine import HipEngine

        eng = HipEngine(model_spec(g.cfg), device)
        eng.configure(g.na, g.nb, g.ns, g.img.shape[1:])
        eng.pack({k: v.to(device) for k, v in g.params.items()})
        flat = FlatParams(mo.param_shapes(g.cfg), device)
        flat.load(g.params)
        fa = FusedA2C(eng, flat, 1e-3, g.gamma, use_graph=mode == "graph")
        for it in range(7):
            if it == 4:
                fa.lr = 5e-4  # changes mid-run: replays must pick it up
            eager = mode == "eager" or it == 3  # one eager iteration between replays
            if eager:
                fa.iteration(img, y, draw_episode_device(eng, 77, it))
            else:
                fa.iteration_graph(img, y, 77, it)
        th.cuda.synchronize()
        return flat.params.clone(), flat.step

    (p0, s0), (p1, s1) = run("eager"), run("graph")
    assert s0 == s1 == 7
    assert (p0 - p1).abs().max().item() <= 2e-6 * p0.abs().max().item()


# ---- input pipeline: decode rate of the worker processes -------------------------------------------
def test_png_folder_loader_rate(device, tmp_path):
    """Trains from a generated PNG folder at 256 x 256 and measures what the loader sustains
    (images / s decoded + uploaded, one rank) next to what one training step consumes; the
    numbers go to gpurun_out/r06_loader.json (DESIGN.md quotes them)."""
    import numpy as np
    from PIL import Image
    from torch.utils.data import DataLoader

    from marlclassification_amd.data import DevicePrefetcher, ImageFolderU8
    from marlclassification_amd.train import ShardedBatchSampler, loader_workers

    rng = np.random.default_rng(0)
    n_img, size = 2048, 256
    for c in range(3):
        os.makedirs(tmp_path / f"class{c}")
    base = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
    for i in range(n_img):
        arr = np.roll(base, i, axis=0)  # distinct but cheap to generate
        Image.fromarray(arr).save(tmp_path / f"class{i % 3}" / f"{i:05d}.png", compress_level=1)
    ds = ImageFolderU8(str(tmp_path), img_size=size)
    assert len(ds) == n_img
    rec = {}
    for workers in (0, loader_workers(ds), min(32, (os.cpu_count() or 1) - 1), min(96, (os.cpu_count() or 1) - 1)):
        bs = ShardedBatchSampler(range(n_img), 64, 0, 1, shuffle=True, seed=1)  # (a worker decodes whole batches)
        dl = DataLoader(ds, batch_sampler=bs, num_workers=workers, pin_memory=True,
                        persistent_workers=workers > 0, prefetch_factor=4 if workers > 0 else None)
        pf = DevicePrefetcher(dl, device)
        for _ in pf:  # warm-up epoch: worker start-up, page cache
            pass
        th.cuda.synchronize()
        t0 = time.perf_counter()
        seen = 0
        for x, yb in pf:
            seen += x.shape[0]
            assert x.dtype == th.uint8 and x.shape[1:] == (3, size, size) and x.is_cuda
        th.cuda.synchronize()
        rec[f"workers_{workers}"] = round(seen / (time.perf_counter() - t0), 1)
        assert seen == n_img
    # the same workers behind several DataLoaders (one collate / pin thread each): data.StripedLoader
    from marlclassification_amd.data import StripedLoader
    for stripes in (2, 4, 8):
        bs = ShardedBatchSampler(range(n_img), 64, 0, 1, shuffle=True, seed=1)
        pf = DevicePrefetcher(StripedLoader(ds, bs, min(64, (os.cpu_count() or 1) - 1), stripes), device)
        seen0 = [y0.clone() for _, y0 in pf]  # warm-up epoch; labels in order
        th.cuda.synchronize()
        t0 = time.perf_counter()
        seen = [y0.clone() for x, y0 in pf]
        th.cuda.synchronize()
        rec[f"striped_{stripes}x_workers_64"] = round(n_img / (time.perf_counter() - t0), 1)
        want = [th.tensor([ds[i][1] for i in b]) for b in bs]
        assert len(seen) == len(want) and all(th.equal(a.cpu(), b) for a, b in zip(seen, want)), "batch order"
    # decode once, keep the uint8 set in HBM: fill cost, then the per-epoch gather rate
    from marlclassification_amd.data import ResidentLoader
    bs = ShardedBatchSampler(range(n_img), 256, 0, 1, shuffle=True, seed=1)
    rl = ResidentLoader(ds, range(n_img), bs, device, workers=min(32, (os.cpu_count() or 1) - 1))
    th.cuda.synchronize()
    t0 = time.perf_counter()
    first = [(x[:1].clone(), y.clone()) for x, y in rl]
    th.cuda.synchronize()
    rec["resident_fill_img_s"] = round(n_img / (time.perf_counter() - t0), 1)  # incl. starting 32 worker processes
    rec["resident_fill_steady_img_s"] = round(rl.fill_img_s, 1)  # from the first decoded chunk on
    t0 = time.perf_counter()
    reps = 20
    for e in range(reps):
        bs.set_epoch(e + 1)
        for x, yb in rl:
            assert x.dtype == th.uint8 and x.shape == (256, 3, size, size) and x.is_cuda
    th.cuda.synchronize()
    rec["resident_img_s"] = round(reps * n_img / (time.perf_counter() - t0), 1)
    # the resident rows are the decoded files: check one batch against a direct decode
    bs.set_epoch(0)
    idx0 = next(iter(bs))
    assert th.equal(first[0][0][0].cpu(), ds[idx0[0]][0]) and first[0][1][0].item() == ds[idx0[0]][1].item()
    rec["cpu_count"] = os.cpu_count()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_loader.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    # more decode processes must not be slower than the training thread decoding alone
    assert max(v for k, v in rec.items() if k.startswith("workers_") and k != "workers_0") >= 0.8 * rec["workers_0"]


# ---- cnn_fwd3 (AidCnn) against the general fused kernel: ragged groups, uint8 pixels --------------------
@pytest.mark.parametrize("f, shape", [(24, (3, 96, 96)), (32, (3, 128, 128))])
@pytest.mark.parametrize("u8", [False, True])
def test_aid_forward_kernel_equals_the_general_kernel(device, f, shape, u8):
    """The workgroup-per-4-patches AidCnn kernel on R = 3 x 3 = 9 patches (a ragged last group) and on
    R = 4 x 5 = 20, float and uint8 images: same features, saved conv outputs and GroupNorm
    statistics as cnn_fwd_kernel (knob cnn_fwd3 = 0), and both within 1e-5 of the oracle's features."""
    from marlclassification_amd import engine as E
    from marlclassification_amd.engine import HipEngine
    from tests.util import model_spec, uniform_params

    cfg = mo.OracleConfig("aid", f, 32, 32, 8, 12, 8, 5, 48, 48, actions=[[3, 0], [-3, 0], [0, 3], [0, -3]])
    params = uniform_params(cfg, 5)
    for na, nb in ((3, 3), (4, 5)):
        ns = 2
        gen = th.Generator().manual_seed(21)
        img = th.randint(0, 256, (nb, *shape), dtype=th.uint8, generator=gen) if u8 else th.rand(nb, *shape, generator=gen)
        inp = mo.draw_episode_inputs(cfg, na, nb, ns, shape[1:], 13)
        got = {}
        for knob in (1, 0):
            E.tune("cnn_fwd3", knob)
            try:
                eng = HipEngine(model_spec(cfg), device)
                eng.configure(na, nb, ns, shape, img_u8=u8)
                eng.pack({k: v.to(device) for k, v in params.items()})
                args = [t.to(device) for t in (img, inp.pos0, inp.h0, inp.c0, inp.hc0, inp.cc0, inp.q)]
                out = eng.episode_forward(*args, None, True)
                got[knob] = (out.step_preds.clone(), [eng.debug_buffer("U", t)[:, : cfg.nf].clone() for t in range(ns)],
                             out.step_pos.clone())
            finally:
                E.tune("cnn_fwd3", 1)
        assert th.equal(got[1][2], got[0][2])
        for a, b in zip(got[1][1], got[0][1]):
            assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())
        assert (got[1][0] - got[0][0]).abs().max().item() <= 1e-5
        ref = mo.run_episode(params, cfg, img.float() / 255 if u8 else img, inp, ns)
        for t in range(ns):
            u = ref.step_u[t].reshape(na * nb, -1)[:, : cfg.nf]
            assert (got[1][1][t].cpu() - u).abs().max().item() <= 1e-5
