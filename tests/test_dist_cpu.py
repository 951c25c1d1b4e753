"""CPU, gloo, world_size 2 and 3: the data-parallel exchange primitives (dense range all-reduce, row-sparse embedding
exchange) against the dense sum they must equal."""
import os

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

import smtc_amd  # noqa: F401
from smtc_amd import dist as mmdist


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    mmdist.init_from_env(backend="gloo")
    assert mmdist.world_size() == world and mmdist.rank() == rank
    g = torch.Generator().manual_seed(100 + rank)
    V, H, n = 50, 8, 24
    ids = torch.randint(0, V, (4, 6), generator=g)
    table = torch.zeros(V, H)
    rows = torch.randn(n, H, generator=g)
    table.index_add_(0, ids.reshape(-1), rows)          # this rank's (row-sparse) gradient
    flat = torch.randn(40, generator=g)
    dense_in = [table.clone(), flat.clone()]
    w = mmdist.allreduce_range(flat, 8, 24)
    mmdist.sparse_rows_exchange(table, ids)
    w.wait()
    # reference: plain dense all-reduce of the same inputs
    ref_table, ref_flat = dense_in[0].clone(), dense_in[1].clone()
    td.all_reduce(ref_table)
    part = ref_flat[8:24].clone()
    td.all_reduce(part)
    ref_flat[8:24] = part
    # ranks with DIFFERENT token counts (each rank trims its batch to its own longest post): fixed number of id slots, the row
    # flags of include/mmhip.h get bit0 on the rows that arrive from the other rank
    n2 = 10 + 6 * rank
    ids2 = torch.randint(0, V, (n2,), generator=g)
    table2 = torch.zeros(V, H)
    table2.index_add_(0, ids2, torch.randn(n2, H, generator=g))
    ref2 = table2.clone()
    td.all_reduce(ref2)
    state = torch.zeros(V + 2, dtype=torch.uint8)
    state[ids2] = 1
    st = mmdist.sparse_rows_exchange_begin(table2, ids2, capacity=24)
    mmdist.sparse_rows_exchange_finish(st, table2, state)
    # replicas must stay BIT-identical (nothing re-synchronises parameters): every rank's summed table equals rank 0's,
    # which "own rows first, then the others" does not give for three or more ranks (fp32 sums do not associate)
    same = True
    for tab in (table, table2):
        gathered = [torch.empty_like(tab) for _ in range(world)]
        td.all_gather(gathered, tab)
        same = same and all(torch.equal(gathered[0], g_) for g_ in gathered[1:])
    # the merged stage buckets equal per-range all-reduces
    flat2 = torch.randn(64, generator=torch.Generator().manual_seed(7 + rank))
    ref3 = flat2.clone()
    td.all_reduce(ref3)
    old = mmdist.BUCKET_BYTES
    mmdist.BUCKET_BYTES = 64                     # 16 floats per bucket
    bk = mmdist.StageBuckets(flat2)
    for b_, e_ in ((0, 8), (8, 20), (20, 20), (20, 30), (30, 64)):
        bk.add(b_, e_, flush=(e_ == 64))
    for w_ in bk.works:
        w_.wait()
    mmdist.BUCKET_BYTES = old
    same = same and bool((flat2 - ref3).abs().max() < 1e-5) and bk.bytes == 64 * 4 and len(bk.works) == 2
    # sharded evaluation: rank r evaluates items r, r + W, ...; the gathered dict is in data-set order on every rank
    import numpy as np

    class _DS(torch.utils.data.Dataset):
        def __len__(self):
            return 11

        def __getitem__(self, i):
            return {"x": torch.tensor(i)}

    loader = torch.utils.data.DataLoader(_DS(), batch_size=4, shuffle=False)
    sh = mmdist.shard_eval_loader(loader)
    mine = np.concatenate([b["x"].numpy() for b in sh])
    full = mmdist.gather_eval({"data_id": mine + 100, "predictions": mine * 2, "labels": mine % 3, "batch_losses": [float(rank + 1)] * 2, "loss": 0.0})
    same = same and full["predictions"].tolist() == [2 * i for i in range(11)] and full["data_id"].tolist() == [100 + i for i in range(11)] \
        and full["labels"].tolist() == [i % 3 for i in range(11)] and abs(full["loss"] - (sum(range(1, world + 1)) / world)) < 1e-9
    touched = (ref2.abs().sum(1) > 0)
    ok_flags = bool(((state[:V] & 1).bool() | ~touched).all())
    q.put((rank, float((table - ref_table).abs().max()), float(max((flat - ref_flat).abs().max(), (table2 - ref2).abs().max())) + (0.0 if ok_flags else 1.0) + (0.0 if same else 2.0)))
    td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_equals_dense_allreduce(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 * world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, e1, e2 in res:
        assert e1 < 1e-5 and e2 < 1e-6


def _shard_worker(rank, world, port, q):
    """dist.ShardedBuckets on CPU tensors: reduce-scatter -> update of the own shard -> all-gather must leave every replica with the parameters
    the all-reduce + replicated update gives (same update rule, here p -= 0.1 * mean gradient), bit-identical across ranks, every element of a
    bucket owned by exactly one rank (or by all: the tail), the non-owned gradient cleared by the caller"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    mmdist.init_from_env(backend="gloo")
    n = 203
    g = torch.Generator().manual_seed(5 + rank)
    grad = torch.randn(n, generator=g)
    param = torch.arange(n, dtype=torch.float32) / 7.0            # identical on every rank
    ref_g = grad.clone()
    td.all_reduce(ref_g)
    ref_p = param - 0.1 * ref_g / world
    old = mmdist.BUCKET_BYTES
    mmdist.BUCKET_BYTES = 200                    # 50 floats per bucket
    bk = mmdist.ShardedBuckets(grad)
    stages = ((0, 28), (28, 64), (64, 64), (64, 131), (131, 203))
    for b_, e_ in stages:
        bk.add(b_, e_, flush=(e_ == n))
    mmdist.BUCKET_BYTES = old
    owned = torch.zeros(n, dtype=torch.int32)
    mom = mmdist.ShardMoments("cpu")
    for ob, oe, replicated in bk.own_ranges():
        param[ob:oe] -= 0.1 * grad[ob:oe] / world
        owned[ob:oe] += 1
        m_, v_ = mom.get(ob, oe)
        assert m_.numel() == oe - ob
    bk.gather_params(param)
    td.all_reduce(owned)
    gathered = [torch.empty_like(param) for _ in range(world)]
    td.all_gather(gathered, param)
    same = all(torch.equal(gathered[0], g_) for g_ in gathered[1:])
    expect = torch.zeros(n, dtype=torch.int32)        # every element updated by exactly one rank -- or, the few tail elements, once on every rank
    for b_, s_, e_, _, _ in bk.plan:
        expect[b_: b_ + world * s_] = 1
        expect[b_ + world * s_: e_] = world
    covered = bool(torch.equal(owned, expect)) and int((expect == world).sum()) < 4 * world * len(bk.plan)
    err = float((param - ref_p).abs().max())
    frac = mom.numel() / n                            # a rank keeps moments for about 1 / world of the elements
    q.put((rank, err, same, covered, frac, len(bk.plan), bk.bytes))
    td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_exchange_equals_allreduce(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 11 * world) % 2000
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, err, same, covered, frac, nplan, nbytes in res:
        assert err < 1e-5 and same and covered and nbytes == 203 * 4 and nplan >= 2
        assert frac < 1.0 / world + 0.25, frac


def _guard_worker(rank, world, port, q):
    """dist.check_shared_device: two ranks that report the same physical GPU and GPU_MAX_HW_QUEUES > 4 is the combination that deadlocked in round 4
    (profiles/r04_hw_queues.txt) -- refused unless overridden; the default queue count, or a GPU per rank, passes"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
    os.environ.pop("MMHIP_ALLOW_SHARED_HW_QUEUES", None)
    mmdist.init_from_env(backend="gloo")
    out = []
    real = mmdist._physical_device
    mmdist._physical_device = lambda: ("host", 0, 5, 0)              # both ranks on one card
    out.append(mmdist.check_shared_device())                         # default queue count: fine, 2 ranks share
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    try:
        mmdist.check_shared_device()
        out.append("no error")
    except RuntimeError as exc:
        out.append("refused" if "GPU_MAX_HW_QUEUES=8" in str(exc) else str(exc))
    os.environ["MMHIP_ALLOW_SHARED_HW_QUEUES"] = "1"
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out.append(mmdist.check_shared_device())
        out.append(len(w))
    os.environ.pop("MMHIP_ALLOW_SHARED_HW_QUEUES")
    mmdist._physical_device = lambda: ("host", 0, 5 + rank, 0)       # a card per rank: 8 queues are the user's business
    out.append(mmdist.check_shared_device())
    mmdist._physical_device = real
    # sharded optimizer: the non-finite counter of a shard's owner reaches every rank (dist.sync_guard_counter) -- here rank 1 "poisons" its shard
    words = torch.tensor([5 * rank, 0], dtype=torch.int32)
    mmdist.sync_guard_counter(words)
    out.append(words.tolist())
    q.put((rank, out))
    td.destroy_process_group()


def test_shared_device_with_many_hardware_queues_is_refused():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out in res:
        assert out == [2, "refused", 2, 1, 1, [5, 0]], out
