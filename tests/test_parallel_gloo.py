"""CPU, world_size 2 over gloo: the data-parallel exchange of parallel.py - batch sharding,
the flat gradient all-reduce (averaged gradient == mean of shard gradients), and the
3-double (n, sum, sum^2) exchange that makes ``standardize`` global.  Gradients come from
the oracle (the HIP kernels need a GPU); the collective logic is what is under test."""
import os
import socket

import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import marl_oracle as mo
from tests.util import Golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_inputs(g, lo, hi):
    i = g.inp
    inp = mo.EpisodeInputs(i.pos0[:, lo:hi], i.h0[:, lo:hi], i.c0[:, lo:hi], i.hc0[:, lo:hi],
                           i.cc0[:, lo:hi], i.q[:, :, lo:hi])
    return g.img[lo:hi], g.y[lo:hi], inp


def _flat(grads, names):
    return th.cat([grads[k].flatten() for k in names])


def _worker(rank, world, port, exact, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    th.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marlclassification_amd.parallel import GradAllReduce, allreduce_adv_stats, shard_bounds

    g = Golden("g1_conftest")
    nb = 18  # divisible by 2 (the fixture holds 19 images)
    lo, hi = shard_bounds(nb, rank, world)
    img, y, inp = _shard_inputs(g, lo, hi)
    names = list(g.params)
    stats = None
    if exact:
        with th.no_grad():
            tr = mo.run_episode(g.params, g.cfg, img, inp, g.ns)
            adv = mo.advantages(tr.step_preds, tr.step_values, y, g.gamma).double()
        st = th.tensor([adv.numel(), adv.sum().item(), (adv * adv).sum().item()], dtype=th.float64)
        allreduce_adv_stats(st)
        n, s1, s2 = st.tolist()
        mean = s1 / n
        stats = (float(mean), float(((s2 - s1 * mean) / (n - 1)) ** 0.5))
    _, _, grads = mo.train_iteration(g.params, g.cfg, img, y, inp, g.ns, g.gamma, adv_stats=stats)
    flat = _flat(grads, names)
    scale = GradAllReduce(world)(flat)
    if rank == 0:
        out_q.put((flat * scale).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exact", [False, True])
def test_two_rank_gradient_exchange(exact):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, exact, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = th.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    g = Golden("g1_conftest")
    names = list(g.params)
    if exact:
        # global statistics -> the averaged gradient IS the big-batch gradient
        img, y, inp = _shard_inputs(g, 0, 18)
        _, _, grads = mo.train_iteration(g.params, g.cfg, img, y, inp, g.ns, g.gamma)
        ref = _flat(grads, names)
    else:
        # per-shard statistics (what DDP on the reference would do): mean of shard gradients
        parts = []
        for lo, hi in ((0, 9), (9, 18)):
            img, y, inp = _shard_inputs(g, lo, hi)
            _, _, grads = mo.train_iteration(g.params, g.cfg, img, y, inp, g.ns, g.gamma)
            parts.append(_flat(grads, names))
        ref = (parts[0] + parts[1]) / 2
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_shard_bounds_and_seeds():
    from marlclassification_amd.parallel import shard_bounds, shard_seed

    assert [shard_bounds(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    with pytest.raises(ValueError):
        shard_bounds(10, 0, 4)
    assert len({shard_seed(42, r) for r in range(8)}) == 8


# ---- input pipeline: index-sharded sampler and the decode-once resident image set -----------------
def _resident_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    th.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marlclassification_amd.data import ResidentLoader, SyntheticImages
    from marlclassification_amd.train import ShardedBatchSampler

    ds = SyntheticImages(37, 3, 8, 5, seed=3)
    part = th.randperm(37, generator=th.Generator().manual_seed(1))[:31].tolist()  # odd: padded shards
    bs = ShardedBatchSampler(part, 8, rank, world, shuffle=True, seed=2)
    rl = ResidentLoader(ds, part, bs, "cpu", workers=0, rank=rank, world=world, chunk=4)
    got = []
    for e in range(2):
        bs.set_epoch(e)
        got.append([(x.numpy().copy(), y.numpy().copy()) for x, y in rl])
    out_q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_resident_image_set_equals_the_streamed_batches():
    """Each rank decodes half of the images, the all-gather rebuilds the set, and the batches a
    rank then gathers are exactly the images its index slices name (two epochs, two shuffles)."""
    from marlclassification_amd.data import SyntheticImages
    from marlclassification_amd.train import ShardedBatchSampler

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_resident_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ds = SyntheticImages(37, 3, 8, 5, seed=3)
    part = th.randperm(37, generator=th.Generator().manual_seed(1))[:31].tolist()
    for rank in range(2):
        bs = ShardedBatchSampler(part, 8, rank, 2, shuffle=True, seed=2)
        for e in range(2):
            bs.set_epoch(e)
            want = list(bs)
            assert len(want) == len(res[rank][e]) == len(bs)
            for idx, (x, y) in zip(want, res[rank][e]):
                assert th.equal(th.from_numpy(x), ds.x[idx]) and th.equal(th.from_numpy(y), ds.y[idx])
    # the two ranks' slices of one global batch are disjoint and cover it
    a = ShardedBatchSampler(part, 8, 0, 2, True, 2)
    b = ShardedBatchSampler(part, 8, 1, 2, True, 2)
    for ia, ib in zip(a, b):
        assert not set(ia) & set(ib) and len(ia) == len(ib)


def _bucket_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marlclassification_amd.fused import FlatParams
    from marlclassification_amd.parallel import HEAD_PREFIXES, BucketedGradAllReduce, GradAllReduce

    g = Golden("g1_conftest")
    flat = FlatParams({k: tuple(v.shape) for k, v in g.params.items()}, th.device("cpu"))
    gen = th.Generator().manual_seed(100 + rank)
    grads = th.randn(flat.numel, generator=gen)
    one = grads.clone()
    scale1 = GradAllReduce(world)(one)
    ar = BucketedGradAllReduce(world, None, flat.offsets, flat.numel, th.device("cpu"))
    two = grads.clone()
    ar.before_backward(None)  # (CPU: no event to install; arms the two-collective form)
    scale2 = ar(two)
    ar.after_backward(None)
    first_head = min(off for k, off in flat.offsets.items() if k.startswith(HEAD_PREFIXES))
    # a layout whose heads are NOT the tail degrades to the single collective
    shuffled = dict(reversed(list(flat.offsets.items())))
    rev = FlatParams({k: tuple(g.params[k].shape) for k in shuffled}, th.device("cpu"))
    if rank == 0:
        out_q.put((bool(th.equal(one, two)), scale1 == scale2, ar.split == first_head and 0 < ar.split < flat.numel,
                   BucketedGradAllReduce(world, None, rev.offsets, rev.numel, th.device("cpu")).split is None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_bucket_gradient_exchange_equals_one_bucket():
    """the heads' slice of the flat gradient buffer (the tail: policy, critic, prediction) and the
    rest as two collectives give the same sums as one (world size 2, gloo, CPU); the split point is the first head
    parameter; a layout with the heads elsewhere falls back to the single all-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, scale_same, split_ok, fallback = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same and scale_same and split_ok and fallback
