"""Data parallelism for the late-fusion step: one process per GPU (torchrun), torch.distributed backend "nccl" (= RCCL
over xGMI on ROCm); "gloo" on CPU for the tests.  The reference has no distributed layer: semantics are
"the reference run per rank on its own B posts (ITC / ITM stay rank-local), gradients averaged".

Exchange plan:
  * dense all-reduce (sum; AdamW multiplies by 1/world) of each backward stage's parameter range, launched right after
    the stage is enqueued, so RCCL overlaps with the remaining backward stages (heads -> layer L-1 ... 0 -> embeddings);
  * the word-embedding gradient (vocab x H, 768 MB for Bernice) is row-sparse: ranks exchange (row id, row) for the rows
    they touched (<= B*T of them) with one all_gather, then add the other ranks' rows locally -- exact, 25 MB per rank
    instead of 768 MB.
"""
import os

import torch
import torch.distributed as td


def world_size():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def init_from_env(backend=None):
    """torchrun / torch.distributed.run environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or (td.is_available() and td.is_initialized()):
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = os.environ.get("MMHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    td.init_process_group(backend=backend)
    check_shared_device()


def _physical_device():
    """(host, PCI location) of the GPU this rank drives, or None without one"""
    if not torch.cuda.is_available():
        return None
    import socket
    p = torch.cuda.get_device_properties(torch.cuda.current_device())
    loc = tuple(getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if all(v is None for v in loc):
        loc = (os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")), torch.cuda.current_device())
    return (socket.gethostname(),) + loc


def check_shared_device(max_queues=4):
    """Two ranks on ONE GPU (the gloo rehearsal of the N > 1 path on a one-GPU box) with GPU_MAX_HW_QUEUES above ROCm's default of 4 deadlocked in
    round 4 (profiles/r04_hw_queues.txt, r05_two_ranks_hw_queues.txt): every process then maps its streams onto up to 8 hardware queues PER PRIORITY
    LEVEL (the engine's side streams, torch's pool, and the high-priority pool streams gloo's device-tensor collectives copy on), two processes ask
    for more hardware queues than the device has slots for, and the step's cross-queue event waits stop making progress once queues are
    time-sliced.  One GPU per rank -- production -- never gets there.  So: refuse that combination (MMHIP_ALLOW_SHARED_HW_QUEUES=1 overrides, for
    the one confirming run).  Returns the number of ranks sharing this rank's device."""
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() < 2:
        return 1
    mine = _physical_device()
    all_ = [None] * td.get_world_size()
    td.all_gather_object(all_, mine)
    sharing = sum(1 for d in all_ if d is not None and d == mine)
    q = os.environ.get("GPU_MAX_HW_QUEUES")
    if sharing > 1 and q is not None and q.isdigit() and int(q) > max_queues:
        msg = (f"{sharing} ranks share GPU {mine} and GPU_MAX_HW_QUEUES={q} (> {max_queues}): this combination deadlocked in the step's first collective wait "
               "(DESIGN.md 6); unset GPU_MAX_HW_QUEUES or give every rank its own GPU")
        if os.environ.get("MMHIP_ALLOW_SHARED_HW_QUEUES", "0") != "1":
            raise RuntimeError(msg)
        import warnings
        warnings.warn(msg + " -- MMHIP_ALLOW_SHARED_HW_QUEUES=1: going on")
    return sharing


def sync_guard_counter(guard_words):
    """Sharded optimizer (ShardedBuckets): only the OWNER of a shard runs AdamW over that shard's summed gradient, so the non-finite counter
    (word [0] of include/mmhip.h mmhip_set_guard) of the dense ranges becomes rank-local -- one rank would raise FloatingPointError (bf16) or halve
    its f16 loss scale while the others go on into the next step's collectives: a hang until the timeout, or loss scales drifting apart.  MAX-reduce
    the counter at the end of the step; every rank then polls the same value and takes the same decision in the same step."""
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.all_reduce(guard_words[0:1], op=td.ReduceOp.MAX)


SKIP_EXCHANGE = False      # bench.py only: time the step without its collectives (replicas diverge; never set while training)


def shard_eval_loader(dataloader):
    """rank r's share (items r, r + W, ...) of an evaluation DataLoader, or None when it cannot / need not be sharded.
    MMHIP_EVAL_SHARD=0 keeps the reference behaviour (every rank evaluates everything)."""
    W = world_size()
    ds = getattr(dataloader, "dataset", None)
    if W == 1 or ds is None or os.environ.get("MMHIP_EVAL_SHARD", "1") == "0" or not hasattr(ds, "__len__"):
        return None
    idx = list(range(rank(), len(ds), W))
    return torch.utils.data.DataLoader(torch.utils.data.Subset(ds, idx), batch_size=dataloader.batch_size, shuffle=False,
                                       collate_fn=dataloader.collate_fn, num_workers=dataloader.num_workers)


def gather_eval(res):
    """all ranks' eval dicts (data_id / predictions / labels of interleaved shards, per-batch losses) -> the full dict in data-set order
    on every rank"""
    W = world_size()
    parts = [None] * W
    td.all_gather_object(parts, res)
    import numpy as np
    n = sum(len(p["predictions"]) for p in parts)
    out = {}
    for key in ("data_id", "predictions", "labels"):
        if any(len(p[key]) != len(p["predictions"]) for p in parts):
            out[key] = np.concatenate([p[key] for p in parts]) if parts[0][key] is not None else None
            continue
        full = np.zeros(n, dtype=np.asarray(parts[0][key]).dtype)
        for r, p in enumerate(parts):
            full[r::W] = p[key]
        out[key] = full
    losses = [l for p in parts for l in p["batch_losses"]]
    out["loss"] = float(np.mean(losses)) if losses else float("nan")
    return out


def force_exchange():
    """MMHIP_FORCE_EXCHANGE=1: run the collectives even at world size 1 (exercises the RCCL call pattern on a one-GPU box)"""
    return os.environ.get("MMHIP_FORCE_EXCHANGE", "0") == "1" and td.is_available() and td.is_initialized()


def allreduce_range(flat_grad, begin, end, async_op=True):
    """sum-all-reduce flat_grad[begin:end] in place; returns the Work handle (or None)"""
    if end <= begin:
        return None
    return td.all_reduce(flat_grad[begin:end], op=td.ReduceOp.SUM, async_op=async_op)


def sparse_rows_exchange_begin(table_grad, ids, capacity=None):
    """table_grad [V, H] holds this rank's gradient rows (non-zero only for rows in `ids`).  Starts the exchange (two
    asynchronous all_gathers of a fixed-size payload -- no host sync: rows are sent once per distinct id, first occurrence
    in sorted order, the other slots carry zeros) and returns the state `sparse_rows_exchange_finish` needs, or None.
    `capacity`: number of id slots every rank sends (ranks may hold different token counts -- the prefetcher trims each
    rank's batch to its own longest post); the surplus slots repeat the last id and so carry zero rows."""
    W = world_size()
    if W == 1 and not force_exchange():
        return None
    ids = ids.reshape(-1).to(table_grad.device)
    sorted_ids, _ = torch.sort(ids)
    if capacity is not None and capacity > sorted_ids.numel():
        sorted_ids = torch.cat([sorted_ids, sorted_ids[-1:].expand(capacity - sorted_ids.numel())])
    first = torch.ones_like(sorted_ids, dtype=torch.bool)
    first[1:] = sorted_ids[1:] != sorted_ids[:-1]
    payload = table_grad.index_select(0, sorted_ids) * first.unsqueeze(1).to(table_grad.dtype)
    ids_all = [torch.empty_like(sorted_ids) for _ in range(W)]
    pay_all = [torch.empty_like(payload) for _ in range(W)]
    works = [td.all_gather(ids_all, sorted_ids, async_op=True), td.all_gather(pay_all, payload, async_op=True)]
    return ids_all, pay_all, works, (sorted_ids, payload)


def sparse_rows_exchange_finish(state, table_grad, row_state=None):
    """afterwards table_grad holds the sum over ranks, BIT-IDENTICAL on every rank: the local rows named in this rank's id
    list are cleared and every rank's payload -- the own one included -- is added in rank order 0..W-1, so each row is the
    same fp32 sum ((0 + g_0) + g_1) + ... everywhere (with "own rows first" three or more ranks would round differently and
    the replicas would drift apart; nothing re-synchronises parameters).  `row_state` (uint8 per row, include/mmhip.h:
    mmhip_adamw_rows) gets bit0 set on the rows received."""
    if state is None:
        return
    ids_all, pay_all, works, (own_ids, _own_payload) = state
    for w in works:
        w.wait()
    table_grad.index_fill_(0, own_ids, 0.0)
    me = rank()
    for r in range(len(ids_all)):
        table_grad.index_add_(0, ids_all[r], pay_all[r])
        if row_state is not None and r != me:        # the own rows were flagged by the backward kernel (padding rows are not)
            row_state[ids_all[r]] = row_state[ids_all[r]] | 1


def sparse_rows_exchange(table_grad, ids, row_state=None):
    """begin + finish in one call"""
    sparse_rows_exchange_finish(sparse_rows_exchange_begin(table_grad, ids), table_grad, row_state)


BUCKET_BYTES = int(os.environ.get("MMHIP_BUCKET_MB", "48")) << 20


class StageBuckets:
    """merges the per-stage gradient ranges of one backward pass into all-reduces of >= BUCKET_BYTES: xGMI is
    point-to-point, a ring all-reduce is bound per link, so a few large collectives beat fourteen 28 MB ones.  Stage ranges
    are adjacent in address order (the flat layout follows the backward order), so a bucket is one slice."""

    def __init__(self, flat_grad):
        self.flat, self.begin, self.end, self.works, self.bytes = flat_grad, None, None, [], 0

    def add(self, b, e, flush=False):
        if e > b:
            if self.begin is not None and b != self.end:
                self.flush()                                   # not adjacent: close the open bucket first
            if self.begin is None:
                self.begin = b
            self.end = e
        if self.begin is not None and (flush or (self.end - self.begin) * 4 >= BUCKET_BYTES):
            self.flush()

    def flush(self):
        if self.begin is not None and self.end > self.begin:
            self.works.append(allreduce_range(self.flat, self.begin, self.end))
            self.bytes += (self.end - self.begin) * 4
        self.begin = self.end = None


def _reduce_scatter_inplace(flat, b, s, W):
    """sum over ranks of flat[b : b + W s]; afterwards rank r's shard flat[b + r s : b + (r + 1) s] holds the sum of that shard (the other
    shards are scratch).  RCCL: one in-place reduce-scatter (recvbuff = sendbuff + rank * count); gloo (CPU tests) has no reduce-scatter: an
    all-reduce of the slice gives every rank every shard's sum, of which only the own one is used."""
    if td.get_backend() == "nccl":
        r = rank()
        return td.reduce_scatter_tensor(flat[b + r * s: b + (r + 1) * s], flat[b: b + W * s], op=td.ReduceOp.SUM, async_op=True)
    return td.all_reduce(flat[b: b + W * s], op=td.ReduceOp.SUM, async_op=True)


def _all_gather_inplace(flat, b, s, W):
    """every rank's shard flat[b + r s : b + (r + 1) s] into all ranks' flat[b : b + W s]"""
    r = rank()
    if td.get_backend() == "nccl":
        return td.all_gather_into_tensor(flat[b: b + W * s], flat[b + r * s: b + (r + 1) * s], async_op=True)
    return td.all_gather([flat[b + i * s: b + (i + 1) * s] for i in range(W)], flat[b + r * s: b + (r + 1) * s].clone(), async_op=True)


class ShardedBuckets(StageBuckets):
    """Gradient exchange + optimizer of the dense ranges as reduce-scatter -> AdamW on the rank's own shard -> all-gather (ZeRO stage 1), selectable
    beside the bucketed all-reduce (MMHIP_DP_OPT=shard).  A bucket [b, e) is cut into W equal shards of s elements (s a multiple of 4); the < 4 W
    elements left at its end take a plain all-reduce and a replicated update.  Rank r receives the summed gradient of shard r only and updates
    only those parameters, with moments it alone keeps: 1/W of the optimizer state, of the AdamW traffic (2.9 GB per step and GPU for the
    replicated update of Bernice's dense part) and of its launches' time; the ring moves the same bytes as an all-reduce (reduce-scatter + all-gather
    ARE its two halves).  Replicas stay bit-identical: every parameter has exactly one writer, the others receive its bytes."""

    def __init__(self, flat_grad):
        super().__init__(flat_grad)
        self.plan = []            # (b, s, e, reduce-scatter Work, tail all-reduce Work)

    def flush(self):
        if self.begin is not None and self.end > self.begin:
            b, e, W = self.begin, self.end, world_size()
            s = ((e - b) // (4 * W)) * 4
            rs = _reduce_scatter_inplace(self.flat, b, s, W) if s > 0 else None
            ar = allreduce_range(self.flat, b + W * s, e) if e > b + W * s else None
            self.plan.append((b, s, e, rs, ar))
            self.bytes += (e - b) * 4
        self.begin = self.end = None

    def own_ranges(self, entries=None):
        """[(begin, end, replicated)] of the gradient elements this rank holds the SUM of after the waits: its shard of every bucket (of `entries`,
        a slice of self.plan, when given), and the tails"""
        r, W, out = rank(), world_size(), []
        for b, s, e, rs, ar in (self.plan if entries is None else entries):
            if rs is not None:
                rs.wait()
            if ar is not None:
                ar.wait()
            if s > 0:
                out.append((b + r * s, b + (r + 1) * s, False))
            if e > b + W * s:
                out.append((b + W * s, e, True))
        return out

    def gather_params(self, flat_param, entries=None):
        W = world_size()
        works = [_all_gather_inplace(flat_param, b, s, W) for b, s, e, _, _ in (self.plan if entries is None else entries) if s > 0]
        for w in works:
            w.wait()


class ShardMoments:
    """AdamW moments of the ranges a rank owns under ShardedBuckets, stored compactly (the bucket plan is the same every step: keyed by range)"""

    def __init__(self, device):
        self.device, self.buf = device, {}

    def get(self, b, e):
        k = (b, e)
        if k not in self.buf:
            self.buf[k] = (torch.zeros(e - b, dtype=torch.float32, device=self.device), torch.zeros(e - b, dtype=torch.float32, device=self.device))
        return self.buf[k]

    def numel(self):
        return sum(m.numel() for m, _ in self.buf.values())


def exchange_stage(model, stage, n_stage, use_itc, use_itm, finishers=None, buckets=None):
    """called right after backward stage `stage` was enqueued; returns async Work handles to wait on before AdamW.  With a
    `finishers` list the word-table exchange of the last stage is only started here: the caller runs the appended callable
    after the dense AdamW, so the all_gather travels while the dense parameters are being updated.  With `buckets`
    (StageBuckets) the dense ranges are merged into large all-reduces; its Work handles are in buckets.works."""
    works = []
    b, e = model._stage_ranges[stage]
    if stage < n_stage - 1:
        if buckets is not None:
            buckets.add(b, e)
            return works
        w = allreduce_range(model._flat_grad, b, e)
        if w is not None:
            works.append(w)
        return works
    # embeddings: [LayerNorm, type, position] dense; word table sparse
    word = next(i for i in model._train_params if i["name"].endswith("word_embeddings.weight"))
    if buckets is not None:
        buckets.add(b, word["offset"], flush=True)
    else:
        w = allreduce_range(model._flat_grad, b, word["offset"])
        if w is not None:
            works.append(w)
    V, H = word["shape"]
    table = model._flat_grad[word["offset"]: word["offset"] + V * H].view(V, H)
    cap_b, cap_t = model._capacity                              # the same on every rank (constructor arguments), unlike B*T of a trimmed batch
    # every ITM row is a copy of a row of `ids` (reference models/mm_late.py:389-414): the distinct word rows of the 2B-post
    # text pass are those of the B original posts, so the id list (and the payload) is B*T slots, not 2*B*T
    state = sparse_rows_exchange_begin(table, model._last["ids"], cap_b * cap_t)
    model._last["exchange_bytes"] = (buckets.bytes if buckets is not None else 0) + cap_b * cap_t * (H * 4 + 8)
    finish = lambda: sparse_rows_exchange_finish(state, table, getattr(model, "_word_row_state", None))
    if finishers is None:
        finish()
    else:
        finishers.append(finish)
    return works
