#!/usr/bin/env python3
"""Can the input pipeline feed the step?  (reference models/datasets.py:125-190)

Generates a data-key tree with N JPEG posts (tools/make_dummy_task.py layout, photo-sized images), then measures posts/s of
  (a) the DataLoader alone (normalise + tokenise + JPEG decode in `--workers` worker processes, per-batch tokenisation in the collate),
  (b) DataLoader -> DevicePrefetcher (pinned copies, GPU resize + normalise) -> MMLate_Model.train_step, the path run_mm_late.py trains on,
  (c) the same train_step on one resident batch (what bench.py times),
at BASELINE config 2's shape (bs = 64, T = 128, 224 x 224, 12 + 12 layers).  Prints one JSON line.

  python tools/loader_bench.py [--posts 1024] [--workers 8] [--batch 64] [--layers 12] [--image_px 480x360] [--item_tokenize]
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def make_posts(root, n, size, distinct=1024):
    """photo-like JPEGs (smooth gradients + noise: compress and decode like camera images, unlike pure noise) and tweet-like texts;
    posts beyond `distinct` are hard links to the first ones (decode cost is per post either way)"""
    from PIL import Image
    import make_dummy_task
    run_dir = make_dummy_task.main(root, n, 1)
    data = os.path.join(root, "work", "models", "data", "text-image")
    rng = np.random.RandomState(1)
    w, h = size
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(min(n, distinct)):
        base = np.stack([(xx * rng.uniform(0.2, 1.0) + yy * rng.uniform(0.2, 1.0) + rng.uniform(0, 255)) % 256 for _ in range(3)], -1)
        img = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(data, f"T{1000 + i}.jpg"), quality=85)
    for i in range(distinct, n):
        dst = os.path.join(data, f"T{1000 + i}.jpg")
        if os.path.exists(dst):
            os.remove(dst)
        os.link(os.path.join(data, f"T{1000 + i % distinct}.jpg"), dst)
    return run_dir


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--posts", type=int, default=4096, help="posts per epoch (64 batches of 64 by default: the pipeline-fill latency of an epoch start is then a few per cent, as on the reference's tasks)")
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--image_px", default="480x360")
    ap.add_argument("--item_tokenize", action="store_true")
    ap.add_argument("--no_ring", action="store_true", help="A/B: decoded images through the DataLoader's result queue (round 3) instead of the pinned shared-memory ring")
    ap.add_argument("--epoch_prefetch", action="store_true", help="MMHIP_EPOCH_PREFETCH: a twin loader on which the next epoch is primed while this epoch's tail is trained on")
    ap.add_argument("--resident_last", action="store_true", help="A/B: only MMLate_Model.warm_start() runs before the workers fork; the resident step is not timed")
    args = ap.parse_args()
    import pandas as pd
    import torch
    import smtc_amd  # noqa: F401
    from transformers import AutoTokenizer
    from smtc_amd.datasets import MM_Dataset, BatchTokenizeCollate, worker_init
    from smtc_amd.image_processing import GpuImageProcessor, RawImageCollate, DevicePrefetcher
    from smtc_amd.mm_late import MMLate_Model

    w, h = (int(v) for v in args.image_px.split("x"))
    tmp = tempfile.mkdtemp(prefix="loader_bench_")
    t0 = time.time()
    run_dir = make_posts(tmp, args.posts, (w, h))
    gen_s = time.time() - t0
    tok = AutoTokenizer.from_pretrained(os.path.join(tmp, "BERNICE"))
    df = pd.read_csv(os.path.join(run_dir, "..", "data", "data_key_imgtxt_random.csv"))
    labels = np.eye(4, dtype=np.int64)[np.arange(len(df)) % 4]
    fmt = os.path.join(run_dir, "..", "data", "text-image", "T{}.jpg")
    dev = torch.device("cuda:0")
    torch.set_num_threads(int(os.environ.get("MMHIP_HOST_THREADS", "4")))      # as run_mm_late.py does
    cfg = types.SimpleNamespace(batch_size=args.batch, num_labels=4, use_clip_loss=False, beta_itc=None, use_tim_loss=False, beta_itm=None, max_length=128, dropout=0.05)
    trainer = MMLate_Model(cfg, "bernice", "vit", "attention", arch=dict(layers_txt=args.layers, layers_img=args.layers), seed=0)
    proc = GpuImageProcessor(size=224, device=dev)
    lr, wd, step = 1e-5, 0.00025, 0
    # (c) first, before any worker process exists: the step on a resident batch -- bench.py's synthetic one (uniform ids, no padding)
    from smtc_amd.synthetic import synthetic_batch
    a = trainer.model.arch
    s_ids, s_mask, s_px, s_oh = synthetic_batch(a["vocab"], 4, args.batch, 128, 1, a["txt_kind"], a["pad_id"], False, a["image"], dev)
    K = 20
    res_ps = float("nan")
    if args.resident_last:
        trainer.warm_start()
    else:
        for _ in range(3):
            step += 1
            trainer.train_step(s_ids, s_mask, s_px, s_oh, None, lr, wd, step)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(K):
            step += 1
            trainer.train_step(s_ids, s_mask, s_px, s_oh, None, lr, wd, step)
        torch.cuda.synchronize()
        res_ps = K * args.batch / (time.time() - t)
    ds = MM_Dataset(df.tweet_id.values, df.text.values, labels, tok, 128, fmt, 224, raw_images=True, batch_tokenize=not args.item_tokenize)
    inner = RawImageCollate(proc)
    ring = None
    if args.workers and not args.no_ring:
        from smtc_amd.image_processing import RingCollate, SharedImageRing
        ring = SharedImageRing(args.workers * 4 * (2 if args.epoch_prefetch else 1) + 6, int(1.1 * args.batch * (w * h * 3 + 16)))
        inner = RingCollate(proc, ring)
    collate = inner if args.item_tokenize else BatchTokenizeCollate(tok, 128, inner)
    kw = dict(num_workers=args.workers, collate_fn=collate, drop_last=True)
    if args.workers:
        kw.update(persistent_workers=True, prefetch_factor=4, worker_init_fn=worker_init)
    loader = torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=True, **kw)

    def epoch_loader_only():
        n = 0
        for b in loader:
            n += b["input_ids"].shape[0]
            if "image_slot" in b:
                ring.release(int(b["image_slot"]))
        return n

    epoch_loader_only()                                   # start the workers, warm the page cache
    t = time.time(); n = epoch_loader_only(); loader_ps = n / (time.time() - t)

    feeds, turn = [DevicePrefetcher(loader, dev, proc, depth=3, trim_padding=False, ring=ring)], [0]
    if args.epoch_prefetch and args.workers:
        twin_collate = RingCollate(proc, ring, owner=3) if ring is not None else RawImageCollate(proc)
        kw2 = dict(kw, collate_fn=twin_collate if args.item_tokenize else BatchTokenizeCollate(tok, 128, twin_collate))
        twin = torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=True, **kw2)
        feeds.append(DevicePrefetcher(twin, dev, proc, depth=3, trim_padding=False, ring=ring))
        for b_ in twin:                                   # start the twin's workers as well
            if "image_slot" in b_:
                ring.release(int(b_["image_slot"]))
    lead = args.workers * 4

    def epoch_train():
        nonlocal step
        n = 0
        t_start = time.time()
        cur = turn[0] % len(feeds)
        turn[0] += 1
        nb = len(feeds[cur])
        for it, b in enumerate(feeds[cur]):
            if len(feeds) > 1 and it == max(0, nb - 1 - lead):
                feeds[1 - cur].prime()                    # the next epoch's workers start while this epoch's tail is trained on
            if n == 0:
                fills.append(time.time() - t_start)          # epoch start: the workers decode their first batches while the GPU waits
            ids, mask, px = trainer._unpack(b)
            step += 1
            trainer.train_step(ids, mask, px, b["labels"], None, lr, wd, step)
            n += ids.shape[0]
        torch.cuda.synchronize()
        return n

    fills = []
    import psutil
    me = psutil.Process()

    def cpu_snapshot():
        t_ = me.cpu_times()
        ch = 0.0
        for k in me.children(recursive=True):
            try:
                c = k.cpu_times()
                ch += c.user + c.system
            except psutil.Error:
                pass
        try:
            stat = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        except OSError:
            stat = {}
        return t_.user + t_.system, ch, int(stat.get("throttled_usec", 0))

    epoch_train()
    c0 = cpu_snapshot()
    del fills[:]
    t = time.time(); n = epoch_train() + epoch_train(); el = time.time() - t; e2e_ps = n / el
    steady_ps = (n - 2 * args.batch) / (el - sum(fills))      # the same two epochs without their pipeline-fill waits
    c1 = cpu_snapshot()
    cpu = {"main_process_cores": round((c1[0] - c0[0]) / el, 2), "worker_cores": round((c1[1] - c0[1]) / el, 2),
           "cgroup_throttled_ms": round((c1[2] - c0[2]) / 1e3, 1), "torch_host_threads": torch.get_num_threads()}

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    print(json.dumps({"posts": args.posts, "image_px": args.image_px, "workers": args.workers, "host_cores": cores, "batch": args.batch, "layers": args.layers,
                      "tokenise": "per item" if args.item_tokenize else "per batch (collate)", "image_handoff": "pinned shared-memory ring" if ring is not None else "DataLoader queue",
                      "ring_pinned": bool(ring is not None and ring.pinned), "epoch_prefetch": bool(args.epoch_prefetch),
                      "loader_only_posts_per_s": round(loader_ps, 1), "loader_to_train_step_posts_per_s": round(e2e_ps, 1),
                      "epoch_fill_ms": round(1e3 * sum(fills) / len(fills), 1), "after_fill_posts_per_s": round(steady_ps, 1),
                      "resident_batch_train_step_posts_per_s": round(res_ps, 1), "end_to_end_over_resident": round(e2e_ps / res_ps, 3) if res_ps == res_ps else None,
                      "cpu_during_training": cpu,
                      "bottleneck": ("GPU step" if e2e_ps > 0.9 * res_ps else
                                     "epoch start: the workers decode their first batches while the GPU waits (epoch_fill_ms); after it the GPU step (+ the GPU resize and the H2D copy)"
                                     if steady_ps > 0.9 * res_ps else
                                     "input pipeline: JPEG decode in the workers + the per-batch hand-off of the decoded images to the training process"),
                      "jpeg_generation_s": round(gen_s, 1)}), flush=True)


if __name__ == "__main__":
    main()
