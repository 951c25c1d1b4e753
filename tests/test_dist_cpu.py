"""CPU, gloo, world_size 2: the data-parallel exchange primitives (dense range all-reduce, row-sparse embedding
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
    touched = (ref2.abs().sum(1) > 0)
    ok_flags = bool(((state[:V] & 1).bool() | ~touched).all())
    q.put((rank, float((table - ref_table).abs().max()), float(max((flat - ref_flat).abs().max(), (table2 - ref2).abs().max())) + (0.0 if ok_flags else 1.0)))
    td.destroy_process_group()


def test_exchange_equals_dense_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, e1, e2 in res:
        assert e1 < 1e-5 and e2 < 1e-6
