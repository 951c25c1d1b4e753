#!/usr/bin/env python3
"""Benchmark of the late-fusion fine-tuning step (BASELINE.json metric: posts/sec, Bernice + ViT-B/16, attention fusion,
bs = 64 per GPU, bf16) on N MI355X of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--aux] [--dtype bf16|f16] [--batch 64] [--no-cpu-baseline]

A step = one full training step on one synthetic batch already resident in HBM: ViT forward (frozen), text forward,
heads, fused loss, backward, gradient exchange (N > 1), fused AdamW, 16-bit weight refresh.  N > 1 is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one rank per GPU, RCCL); weak scaling:
every rank has its own 64 posts.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per post, BASELINE.md 2 (2*MAC; forward 58.173, backward text 44.695 + heads 0.930; ViT frozen)
GF_PER_POST = {"plain": 103.798, "aux": 171.528,
               # BASELINE config 4 (CLIP-ViT-L/14 frozen tower + Bernice, concat): 24 x (2 P (4 Hv^2 + 2 Hv Iv) + 4 P^2 Hv) + patch embed,
               # Hv = 1024, Iv = 4096, P = 257 (224 px) / 577 (336 px); + text forward 22.347 + text backward 44.695 + heads 0.01
               "clip224": 229.1, "clip336": 448.9}
STRICT_DTYPE = "bf16x3"     # the dtype whose per-post outputs meet north_star's 1e-3: timed beside the headline as `at_tolerance`
# ... with ONE bf16 MFMA product in the backward's matrix products (include/mmhip.h mmhip_set_backward_products; the forward keeps three): north_star
# states its tolerance on logits and loss, which the backward's product count cannot touch; what it does to the gradients is measured and stated
STRICT_BWD_PRODUCTS = 1
STRICT_GRAD_NOTE = ("forward: three bf16 MFMA products per matrix product (logits, loss: north_star's 1e-3 holds, measured in `parity`); backward: ONE product in the GEMMs and in four "
                    "of the attention backward's five products (the re-computed scores keep three) -- parameter gradients within 1.2e-2 relative L2 per tensor of the fp32 reference "
                    "(measured <= 7.7e-3 at twelve layers, median 4.7e-3, 1 - cos of the flat gradient 8e-6: profiles/r05_x3_bwd_policy.txt; asserted in "
                    "tests/test_gpu_model.py::test_backward_product_policy_keeps_outputs_and_bounds_gradients).  What that costs: the tolerance holds per step on given weights, NOT along a "
                    "training run -- Adam's sign-like early steps turn 5e-3 of gradient noise into 1e-2 of logit drift after 16 steps at the reference's lr (three products: 3e-6; the bf16 / f16 "
                    "modes: 5e-2 after 16, same file).  `strict_backward` in this block times the step with three products everywhere, whose trajectory stays within 6e-5")
PEAK_TFLOPS = 2500.0        # bf16 / f16 dense MFMA, MI355X_MICROARCH.md "Chip-level parameters"
METRIC = {2: "posts/sec (fwd+bwd) Bernice+ViT-B/16 attn-fusion, bs=64",                      # BASELINE.json's metric, quoted on config 2
          3: "posts/sec (fwd+bwd) Bernice+ViT-B/16 attn-fusion + ITC+ITM, bs=64",
          4: "posts/sec (fwd+bwd) CLIP-ViT-L/14 + Bernice concat-fusion, bs=32",
          5: "posts/sec (fwd+bwd) LXMERT early-fusion, bs=32"}


def measure_parity(dtype):
    """the reference's forward goldens (tests/golden/fwd_*.npz: XLM-R and BERT, attention and concat fusion, 2 / 6 / 12 layers, several seeds,
    batch sizes, lengths and padding patterns; vectors produced by the reference's own MM_Model) through the HIP path in THIS run's dtype: max|got - ref| / max|ref| per output, worst over the goldens.  The oracle module is
    used as the checker only (deterministic parameter recipe + synthetic batch of the goldens)."""
    import ast
    import numpy as np
    import torch
    from oracle import mm_oracle as O
    from smtc_amd.mm_late import MM_Model
    worst = {"out_cls": 0.0, "logits_per_text": 0.0, "out_tim": 0.0, "mm_features": 0.0}
    every = {k: [] for k in worst}          # per golden, for the median and the name of the worst one
    gdir = os.path.join(ROOT, "tests", "golden")
    names = sorted(f[:-4] for f in os.listdir(gdir) if f.startswith("fwd_") and f.endswith(".npz"))      # 4 of round 1 + 9 of round 4 (fwd_x_*)
    for name in names:
        z = np.load(os.path.join(gdir, name + ".npz"), allow_pickle=False)
        cfg = O.OracleConfig(**ast.literal_eval(str(z["cfg"])))
        txt = "bert" if cfg.txt_kind == "bert" else "bernice"
        B, T = int(z["B"]), int(z["T"])
        arch = dict(layers_txt=cfg.layers_txt, layers_img=cfg.layers_img, vocab=cfg.vocab, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab,
                    p_hidden=cfg.p_hidden, p_attn=cfg.p_attn)
        m = MM_Model(cfg.num_labels, txt, "vit", cfg.p_head, cfg.fusion, arch=arch, dtype=dtype, max_posts=B, max_text_len=T)
        m.load_state_dict(O.make_params(cfg, int(z["seed_w"])), strict=False)
        m.eval()
        pixels = O.synthetic_batch(cfg, B, T, int(z["seed_x"]), bool(z["pad"]))[2]
        with torch.no_grad():
            o = m(torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]), pixels, tim_inputs=(torch.from_numpy(z["tim_ids"]), torch.from_numpy(z["tim_mask"])))
        for k, v in (("out_cls", o[0]), ("logits_per_text", o[1]), ("out_tim", o[2]), ("mm_features", o[4])):
            ref = torch.from_numpy(z[k])
            err = (v.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
            worst[k] = max(worst[k], err)
            every[k].append((err, name))
        del m
    torch.cuda.empty_cache()
    out = {k: float("%.3g" % v) for k, v in worst.items()}
    out["meets_1e-3"] = all(v < 1e-3 for v in worst.values())
    median = {k: float("%.3g" % sorted(e for e, _ in v)[len(v) // 2]) for k, v in every.items()}
    return {"metric": "max|got-ref|/max|ref| vs the reference's fp32 golden vectors (%d forward goldens, worst), measured in this run" % len(names),
            "north_star_tolerance": 1e-3, "dtype": dtype, "measured": out, "median_over_goldens": median,
            "worst_golden": {k: max(v)[1] for k, v in every.items()},
            "spread_note": "a 16-bit mode's worst case is ONE realisation of its rounding noise: when round 5 replaced the GELU by a form that differs from the old one by "
                           "<= 5e-7, bf16's worst out_cls moved 3.65e-2 -> 5.8e-2 on the same goldens while the parity mode's moved 3.9e-5 -> 2.8e-5; read the median beside it",
            "note": "no single- or two-product 16-bit policy meets 1e-3 on the thirteen goldens (profiles/r04_numerics_study.txt: bf16 3.95e-2, f16 5.1e-3, "
                    "two-product f16 2.5e-3); the dtype that does is bf16x3 (three bf16 products of hi/lo planes: <= 6e-5 here; `at_tolerance` times it in this run)"}


def cpu_baseline(seconds_budget=25.0):
    """the CPU oracle (plain PyTorch fp32 restatement of the reference path) timed on this box's host cores:
    forward+backward of the same model on a bounded sample (B=8 posts of the same synthetic shape)."""
    import torch
    from oracle import mm_oracle as O
    # the box's CPU share, not the host's core count (a GPU box exposes 256 logical CPUs but grants about 16)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(num_labels=2)
    P = {k: v.requires_grad_(O.trainable(k)) for k, v in O.make_params(cfg, 0).items()}
    B = 8
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, 128, 1234, False)
    drop = O.Dropout("torch")

    def step():
        for p in P.values():
            p.grad = None
        out_cls, lpt, _, _, _ = O.mm_forward(P, ids, mask, pixels, cfg, None, drop)
        O.mix_loss(out_cls, onehot, None, lpt, None, None, False, False).backward()

    step()
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < seconds_budget * 0.6 and n < 4):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 3), "unit": "posts/s", "cores": cores, "kind": "port",
            "sample": f"{n} fwd+bwd steps of B={B} posts (T=128, 224x224), fp32 torch CPU oracle, optimizer excluded"}


def bench_early(args):
    """BASELINE config 5 (LXMERT early fusion, mm_early.py): the native early-fusion engine (csrc/early.hip, round 4) -- one C call per step.
    Step = forward (with --aux the ITM pass batched with the main pass: 2B posts) + loss mix + backward + AdamW + operand refresh.  FLOPs per
    post: every Linear of the 9 language / 5 relational / 5 cross-modality layers on T = 128 tokens and 36 boxes, x3 for forward + backward."""
    import types
    import numpy as np
    import torch
    import smtc_amd  # noqa: F401
    from smtc_amd.mm_early import MMEarly_Model
    from smtc_amd import dist as mmdist
    mmdist.init_from_env()
    world, rank = mmdist.world_size(), mmdist.rank()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    torch.set_num_threads(int(os.environ.get("MMHIP_HOST_THREADS", "4")))      # see main(): idle intra-op threads must not throttle the enqueuing thread
    B, T, NB, H, I, C = args.batch, 128, 36, 768, 3072, 3
    cfg = types.SimpleNamespace(batch_size=B, num_labels=C, use_clip_loss=args.aux, beta_itc=0.1, use_tim_loss=args.aux, beta_itm=0.1, max_length=T, dropout=0.05)
    tr = MMEarly_Model(cfg, "lxmert", dtype=args.dtype, seed=0)
    g = torch.Generator().manual_seed(1234 + rank)
    ids = torch.randint(1, 30522, (B, T), generator=g).cuda()
    mask = torch.ones(B, T, dtype=torch.int64).cuda()
    tt = torch.zeros_like(ids)
    feats = (torch.rand(B, NB, 2048, generator=g) * 2).cuda()
    boxes = torch.rand(B, NB, 4, generator=g).cuda()
    onehot = torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=g), C).cuda()
    np.random.seed(30 + rank)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        step += 1
        tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, step)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step += 1
        loss = tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, step)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([el], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        el = float(tmax.item())
    # host time to enqueue one step from an idle queue
    sync()
    th = time.perf_counter()
    for _ in range(2):
        step += 1
        tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, step)
    host_ms = (time.perf_counter() - th) / 2 * 1e3
    sync()
    # roofline of the dominant kernel family (NT GEMMs): HIP events around every launch on the stream it goes to (mmhip_early_gemm_timing),
    # two steps, the engine's three streams on -- the conditions of the timed step
    import ctypes as Ct
    from smtc_amd import _lib
    lib = _lib.lib()
    gemm = None
    if world == 1 and hasattr(lib, "mmhip_early_gemm_timing"):
        _lib.check(lib.mmhip_early_gemm_timing(tr.model._handle, 1, 1, None, None, None))
        for _ in range(2):
            step += 1
            tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, step)
        gms, gl, gf = Ct.c_double(), Ct.c_uint64(), Ct.c_double()
        _lib.check(lib.mmhip_early_gemm_timing(tr.model._handle, 0, 1, Ct.byref(gms), Ct.byref(gl), Ct.byref(gf)))
        if gms.value > 0:
            gemm = {"tflops": gf.value / (gms.value * 1e-3) / 1e12, "ms_per_step": gms.value / 2, "launches_per_step": int(gl.value) // 2,
                    "flops_per_launch": gf.value / max(1, gl.value)}
    layer = 2.0 * (4 * H * H + 2 * H * I)                       # Linear FLOPs per token of a BERT-shaped layer
    xlayer = 2.0 * (8 * H * H + 2 * H * I)                      # cross-modality layer: cross + self attention blocks, feed-forward
    fwd = (9 * layer + 5 * xlayer) * T + (5 * layer + 5 * xlayer) * NB + 2.0 * NB * (2048 + 4) * H
    gf_post = 3 * fwd * (2 if args.aux else 1) / 1e9
    tf = B * args.steps / el * gf_post / 1e3                     # per GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the LXMERT oracle (kind "port") on the box's CPU share: forward + backward on a bounded sample (B = 4 posts of the same shape)
        from oracle import lxmert_oracle as L
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
        torch.set_num_threads(cores)
        lc = L.LxmertConfig(num_labels=C)
        LP = {k: v.requires_grad_(True) for k, v in L.make_params(lc, 0).items()}
        c_ids, c_mask, c_tt, c_feats, c_boxes, c_oh = L.synthetic_batch(lc, 4, T, 1234)

        def cstep():
            for q in LP.values():
                q.grad = None
            o_, et, ev_, ot = L.early_forward(LP, c_ids, c_mask, c_tt, c_feats, c_boxes, lc)
            L.mix_loss(LP, o_, c_oh, None, et, ev_, ot, None, False, False).backward()
        cstep()
        tc, n = time.time(), 0
        while n < 1 or (time.time() - tc < 15.0 and n < 4):
            cstep()
            n += 1
        cpu = {"value": round(4 * n / (time.time() - tc), 3), "unit": "posts/s", "cores": cores, "kind": "port",
               "sample": f"{n} fwd+bwd steps of B=4 posts (T=128, 36 x 2048 ROI features), fp32 torch CPU oracle (oracle/lxmert_oracle.py), optimizer excluded"}
    out = {"metric": METRIC[5] + (" + ITC+ITM" if args.aux else ""), "workload_id": "config5", "value": round(world * B * args.steps / el, 1), "unit": "posts/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "BASELINE config 5: LXMERT early fusion (mm_early.py), 36 x 2048 ROI features, bs=32/GPU" + (", ITC + ITM" if args.aux else ""),
                      "implementation": "native engine (csrc/early.hip: one C call per step; language / vision / weight-gradient streams; per-layer AdamW beside the backward; ITM pass batched with the main pass)",
                      "gf_per_post": round(gf_post, 1), "posts_per_gpu": B, "text_tokens": T, "boxes": NB, "parallelism": f"dp{world}", "weights": "random-init at true shapes"},
           "final_loss": round(float(loss), 5), "host_enqueue_ms_per_step": round(host_ms, 3),
           "roofline": ({"bound": "mfma", "kernel": "NT GEMM family (gemm_nt_kernel 128x128 tiles on 2- / 3-deep LDS rings, MFMA 16x16x32, LDS-DMA staged)",
                         "achieved": round(gemm["tflops"], 1), "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gemm["tflops"] / PEAK_TFLOPS, 4),
                         "conditions": "HIP events around every NT launch on its own stream, the engine's three streams on (as in the timed step): launches of the language, "
                                       "vision and weight-gradient streams share the chip, so the durations add up to more than the step",
                         "gemm_ms_per_step": round(gemm["ms_per_step"], 3), "launches_per_step": gemm["launches_per_step"],
                         "algorithmic_flops_per_launch": round(gemm["flops_per_launch"]), "traffic": None,
                         "traffic_note": "no PMC pass of this command in profiles/ for these kernel sources",
                         "whole_step_tflops": round(tf, 1), "whole_step_frac": round(tf / PEAK_TFLOPS, 4)} if gemm else
                        {"bound": "mfma", "kernel": "whole step: algorithmic Linear FLOPs / step time", "achieved": round(tf, 1), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_TFLOPS, 4), "traffic": None}),
           "cpu_baseline": cpu}
    if world > 1:
        out["multi_gpu"] = "staged exchange: the engine calls back per backward stage, ranges leave as bucketed all-reduces beside the stages below (unmeasured on hardware: the development box has one GPU)"
        torch.distributed.barrier()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--aux", action="store_true", help="BASELINE config 3: ITC + ITM auxiliary losses")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config index (3 = --aux; 4 = CLIP-ViT-L/14 + concat, bs=32; 5 = LXMERT early fusion on the native engine csrc/early.hip, bs=32)")
    ap.add_argument("--image", type=int, default=224, choices=[224, 336], help="config 4: image size (257 / 577 image tokens)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "bf16x3"])
    ap.add_argument("--bwd-products", type=int, default=0, choices=[0, 1, 2, 3],
                    help="--dtype bf16x3 only: bf16 MFMA products per slice in the backward's matrix products (0 = the library default, 3; the forward always takes three)")
    ap.add_argument("--batch", type=int, default=0, help="posts per GPU (default: 64; 32 for config 4)")
    ap.add_argument("--txt_model_name", default="bernice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-at-tolerance", action="store_true", help="skip the second timed loop in the strict-parity dtype")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity measurement against the reference's forward goldens")
    ap.add_argument("--gemm-shapes", default="", help="append the per-shape table of the timed NT GEMM launches to this file")
    args = ap.parse_args()
    if args.config == 3:
        args.aux = True
    if not args.batch:
        args.batch = 32 if args.config in (4, 5) else 64
    if args.config == 5:
        return bench_early(args)
    img_name = "vit" if args.config != 4 else ("clip" if args.image == 224 else "clip336")
    fusion = "concat" if args.config == 4 else "attention"

    import types
    import numpy as np
    import torch
    import smtc_amd  # noqa: F401
    from smtc_amd import _lib, dist as mmdist
    from smtc_amd.mm_late import MMLate_Model
    from smtc_amd.synthetic import synthetic_batch

    mmdist.init_from_env()
    world, rank = mmdist.world_size(), mmdist.rank()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}")
    torch.cuda.set_device(dev)
    # as run_mm_late.py does: torch's default intra-op pool is one thread per HOST core (256 on a GPU box that grants about 16); the
    # idle threads of CPU-side tensor work (weight init, the parity checker's parameter recipe) spin on the box's CPU share beside the
    # thread that enqueues the step (round 3's loader finding).  It was NOT what slowed the strict-dtype loop below after the parity
    # pass -- that was streams sharing a hardware queue -- but the pin stays: the bench then runs like the trainer does
    torch.set_num_threads(int(os.environ.get("MMHIP_HOST_THREADS", "4")))

    B, T, C = args.batch, 128, (3 if args.aux else 2)
    cfg = types.SimpleNamespace(batch_size=B, num_labels=C, use_clip_loss=args.aux, beta_itc=0.1 if args.aux else None,
                                use_tim_loss=args.aux, beta_itm=0.1 if args.aux else None, max_length=T, dropout=0.05)
    kw_bp = {"backward_products": args.bwd_products} if (args.dtype == "bf16x3" and args.bwd_products) else {}
    trainer = MMLate_Model(cfg, args.txt_model_name, img_name, fusion, dtype=args.dtype, seed=0, **kw_bp)
    a = trainer.model.arch
    ids, mask, pixels, onehot = synthetic_batch(a["vocab"], C, B, T, 1234 + rank, a["txt_kind"], a["pad_id"], False, a["image"], dev)
    np.random.seed(30 + rank)
    lr, wd = 1e-5, 0.00025

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step_no = 0
    for _ in range(args.warmup):
        step_no += 1
        trainer.train_step(ids, mask, pixels, onehot, None, lr, wd, step_no)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_no += 1
        loss, _ = trainer.train_step(ids, mask, pixels, onehot, None, lr, wd, step_no)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_step = elapsed / args.steps * 1e3
    posts_s = world * B * args.steps / elapsed
    final_loss = float(loss[0].item())
    # host time to enqueue one step, from an idle queue (inside the timed loop the host runs ahead until the launch queue is
    # full and then waits for the GPU, which says nothing about the host)
    th = time.perf_counter()
    for _ in range(2):
        step_no += 1
        trainer.train_step(ids, mask, pixels, onehot, None, lr, wd, step_no)
    host_ms = (time.perf_counter() - th) / 2 * 1e3
    sync()

    # ---- N > 1: what the gradient exchange costs -- bytes on the wire per rank and step, and the step time it leaves exposed
    # (the same steps timed without the collectives; the replicas diverge, which no longer matters after the timed region)
    exch = None
    if world > 1:
        exch_bytes = int(trainer.model._last.get("exchange_bytes", 0))      # of the last exchanging step (the steps below skip the collectives)
        mmdist.SKIP_EXCHANGE = True
        for _ in range(2):
            step_no += 1
            trainer.train_step(ids, mask, pixels, onehot, None, lr, wd, step_no)
        sync()
        t2 = time.perf_counter()
        nx = max(3, args.steps // 2)
        for _ in range(nx):
            step_no += 1
            trainer.train_step(ids, mask, pixels, onehot, None, lr, wd, step_no)
        sync()
        noex_ms = (time.perf_counter() - t2) / nx * 1e3
        mmdist.SKIP_EXCHANGE = False
        exch = {"exchange_bytes_per_step": exch_bytes, "ms_per_step_without_exchange": round(noex_ms, 3),
                "exposed_exchange_ms": round(ms_step - noex_ms, 3), "hardware_note": "RCCL path measured only where the driver provides > 1 GPU"}

    # ---- extra: forward+backward only (no optimizer / refresh), same batch
    lib, m = _lib.lib(), trainer.model
    sync()
    # phase ends inside a step (HIP events on the phases' own streams, no profiler): image tower | text tower | forward | backward | step
    import ctypes as Ct
    spans = None
    if world == 1 and hasattr(lib, "mmhip_step_spans"):
        acc = []
        _lib.check(lib.mmhip_step_spans(m._handle, 1, None))
        for _ in range(4):
            step_no += 1
            trainer.train_step(ids, mask, pixels, onehot, None, lr, wd, step_no)
            buf = (Ct.c_float * 5)()
            _lib.check(lib.mmhip_step_spans(m._handle, 1, buf))
            acc.append(list(buf))
        _lib.check(lib.mmhip_step_spans(m._handle, 0, None))
        acc = sorted(acc[1:], key=lambda r: r[4])[len(acc[1:]) // 2]
        spans = dict(zip(["image_tower_end", "text_tower_end", "forward_end", "backward_end", "step_end"], (round(x, 3) for x in acc)))
    sync()
    t1 = time.perf_counter()
    nfb = max(3, args.steps // 4)
    for _ in range(nfb):
        tim = trainer.prepare_itm_inputs(ids, mask) if args.aux else (None, None, None)
        m._engine_forward(ids, mask, pixels, tim[0], tim[1])
        lo = torch.empty(4, device=dev)
        w_cls, w_itc, w_itm = trainer.loss_weights()
        _lib.check(lib.mmhip_loss(m._handle, _lib.ptr(onehot), None, _lib.ptr(tim[2]), w_cls, w_itc, w_itm, _lib.ptr(lo), None, _lib.stream_ptr()))
        _lib.check(lib.mmhip_backward(m._handle, None, None, None, None, _lib.stream_ptr()))
    torch.cuda.synchronize()
    fb_ms = (time.perf_counter() - t1) / nfb * 1e3
    m._flat_grad.zero_()
    m._word_row_state.bitwise_and_(0xFE)        # include/mmhip.h backward contract: no stale row flags into the next fused step

    # ---- roofline of the dominant kernel (MFMA NT GEMM): HIP events around every launch, on the stream it is launched on.
    # Two passes: side streams ON (the conditions of the timed step: a launch may share the chip with the other tower or with
    # the weight-gradient GEMM -- this is `frac`) and side streams OFF (every kernel alone on the chip: `frac_serial`).
    def gemm_pass(mode):
        nonlocal step_no
        _lib.check(lib.mmhip_gemm_timing(m._handle, mode, 1, None, None, None))
        for _ in range(2):
            step_no += 1
            trainer.train_step(ids, mask, pixels, onehot, None, lr, wd, step_no)
        gms, gl, gf = Ct.c_double(), Ct.c_uint64(), Ct.c_double()
        buf = Ct.create_string_buffer(1 << 16)
        _lib.check(lib.mmhip_gemm_timing_by_shape(m._handle, buf, len(buf)))
        table = buf.value.decode()
        if args.gemm_shapes and rank == 0:
            with open(args.gemm_shapes, "a") as f:
                f.write("# NT GEMM launches by shape, side streams %s (2 steps)\n%s\n" % ("on" if mode == 1 else "off", table))
        # CU-share-weighted time: a launch capped at c workgroups (the forward's CU partition) occupies c of the 256 CUs
        occ_ms = 0.0
        for line in table.splitlines()[1:]:
            f_ = line.split()
            if len(f_) >= 9:
                occ_ms += float(f_[8]) * int(f_[5]) / 256.0
        _lib.check(lib.mmhip_gemm_timing(m._handle, 0, 1, Ct.byref(gms), Ct.byref(gl), Ct.byref(gf)))
        tf = gf.value / (gms.value * 1e-3) / 1e12 if gms.value > 0 else 0.0
        gemm_pass.occ_tf = gf.value / (occ_ms * 1e-3) / 1e12 if occ_ms > 0 else 0.0
        return tf, gms.value, int(gl.value), gf.value

    achieved, gms, gl, gf = gemm_pass(1)
    occ_tf = gemm_pass.occ_tf
    serial_tf, gms_serial, _, _ = gemm_pass(2)
    # HBM bytes per launch of that kernel come from PMC passes of this very command (FETCH_SIZE x2 per the gfx950 correction and
    # WRITE_SIZE, separate rocprofv3 --pmc runs, tools/pmc_traffic.py): bench.py cannot run the profiler on itself, so the
    # number is taken from the committed profile ONLY when that profile was collected on the kernel sources of this build
    # (sha256 over csrc/*.hip, *.h); otherwise it is null
    traffic, traffic_src = None, None
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "socialmedia-textimage-classification-auxlosses_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(csrc, fn), "rb") as f:
                h.update(f.read())
    src_hash = h.hexdigest()[:16]
    for tfile in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_gemm_traffic.json")), reverse=True):
        with open(os.path.join(ROOT, "profiles", tfile)) as f:
            tj = json.load(f)
        if tj.get("csrc_sha256_16") == src_hash and args.config == 2 and not args.aux and B == 64 and world == 1 and tj.get("dtype", "bf16") == args.dtype:
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), "profiles/" + tfile
            break
    roofline = {"bound": "mfma", "kernel": "NT GEMM family (gemm_nt8_kernel / gemm_nt_kernel, MFMA 16x16x32, LDS-DMA staged)",
                "achieved": round(achieved, 1), "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS, 4),
                "conditions": "HIP events around every NT launch on its own stream, side streams on (as in the timed step)",
                "achieved_serial": round(serial_tf, 1), "frac_serial": round(serial_tf / PEAK_TFLOPS, 4),
                "achieved_per_cu_share": round(occ_tf, 1), "frac_per_cu_share": round(occ_tf / PEAK_TFLOPS, 4),
                "cu_share_note": "where the engine partitions the forward (image tower much longer than the text tower: config 4; MMHIP_PART elsewhere) the towers' GEMMs are persistent launches capped at c workgroups = c CUs (DESIGN.md 7c): "
                                 "frac_per_cu_share weighs a launch's duration by the share of the chip it occupies; frac weighs every launch as if it had all 256 CUs",
                "traffic": traffic, "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                "csrc_sha256_16": src_hash,
                "algorithmic_flops_per_launch": round(gf / max(1, gl)),
                "launches_per_step": gl // 2, "avg_launch_us": round(gms * 1e3 / max(1, gl), 2),
                "gemm_ms_per_step": round(gms / 2, 3), "gemm_ms_per_step_serial": round(gms_serial / 2, 3)}
    mode = ("clip224" if args.image == 224 else "clip336") if args.config == 4 else ("aux" if args.aux else "plain")
    parity = None
    if rank == 0 and world == 1 and not args.no_parity:
        parity = measure_parity(args.dtype)
    out = {
        "metric": METRIC[4] if args.config == 4 else (METRIC[3] if args.aux else METRIC[2]),
        "workload_id": "config4" if args.config == 4 else ("config3" if args.aux else "config2"),
        "value": round(posts_s, 1), "unit": "posts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "backward_products": (args.bwd_products or 3) if args.dtype == "bf16x3" else None, "data": "synthetic",
        "config": {"workload": (f"BASELINE config 4: CLIP-ViT-L/14 image encoder ({args.image} px) + Bernice, concat fusion, bs={B}/GPU" if args.config == 4 else
                                "BASELINE config 3: Bernice+ViT-B/16, attention fusion, ITC+ITM, bs=64/GPU" if args.aux else
                                "BASELINE config 2: Bernice+ViT-B/16, attention fusion, no aux loss, bs=64/GPU"),
                   "gf_per_post": GF_PER_POST[mode],
                   "step": "full train step: fwd + loss + bwd + grad exchange + AdamW + weight refresh",
                   "posts_per_gpu": B, "text_tokens": T, "image": a["image"], "vocab": a["vocab"], "parallelism": f"dp{world}",
                   "weights": "random-init at true shapes"},
        "host_enqueue_ms_per_step": round(host_ms, 3), "fwd_bwd_ms": round(fb_ms, 3), "fwd_bwd_posts_per_s": round(world * B / (fb_ms * 1e-3), 1),
        "model_tflops": round(posts_s * GF_PER_POST[mode] / 1e3, 1),
        "model_frac_of_peak": round(posts_s / world * GF_PER_POST[mode] / 1e3 / PEAK_TFLOPS, 4),
        "final_loss": round(final_loss, 5), "spans_ms": spans, "roofline": roofline,
        "parity": parity,
    }
    if exch is not None:
        out["exchange"] = exch
    # ---- the same step in the dtype that meets north_star's 1e-3 on the per-post outputs, timed in THIS run (VERDICT r3 #1a): the
    # headline dtype is the throughput mode; `at_tolerance` is the throughput that satisfies the parity bar
    if rank == 0 and world == 1 and not args.no_at_tolerance and args.config in (2, 3):
        if parity is not None and parity["measured"]["meets_1e-3"]:
            out["at_tolerance"] = {"dtype": args.dtype, "ms_per_step": round(ms_step, 3), "posts_per_s": round(posts_s, 1),
                                   "model_frac_of_peak": out["model_frac_of_peak"], "parity": parity["measured"], "meets_1e-3": True,
                                   "note": "the headline dtype itself meets the tolerance"}
        else:
            del trainer, m
            torch.cuda.empty_cache()

            def strict_loop(products):
                t2 = MMLate_Model(cfg, args.txt_model_name, img_name, fusion, dtype=STRICT_DTYPE, seed=0, backward_products=products)
                np.random.seed(30 + rank)
                sn = 0
                for _ in range(min(args.warmup, 3)):
                    sn += 1
                    t2.train_step(ids, mask, pixels, onehot, None, lr, wd, sn)
                sync()
                ks = max(3, min(args.steps, 10))
                ta = time.perf_counter()
                marks = []
                for _ in range(ks):
                    sn += 1
                    t2.train_step(ids, mask, pixels, onehot, None, lr, wd, sn)
                    if os.environ.get("BENCH_AT_TRACE"):
                        sync(); marks.append(time.perf_counter())
                sync()
                ms2 = (time.perf_counter() - ta) / ks * 1e3
                if marks:
                    print("at_tolerance per-step ms:", " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip([ta] + marks[:-1], marks)), file=sys.stderr, flush=True)
                del t2
                torch.cuda.empty_cache()
                return ks, ms2
            ks, ms2 = strict_loop(STRICT_BWD_PRODUCTS)
            ks3, ms3 = strict_loop(3)
            p2 = measure_parity(STRICT_DTYPE)["measured"] if not args.no_parity else None
            frac = lambda ms: round(B / (ms * 1e-3) * GF_PER_POST[mode] / 1e3 / PEAK_TFLOPS, 4)
            out["at_tolerance"] = {"dtype": STRICT_DTYPE, "forward_products": 3, "backward_products": STRICT_BWD_PRODUCTS, "gradients": STRICT_GRAD_NOTE,
                                   "steps": ks, "ms_per_step": round(ms2, 3), "posts_per_s": round(B / (ms2 * 1e-3), 1),
                                   "model_frac_of_peak": frac(ms2),
                                   "slowdown_vs_headline": round(ms2 / ms_step, 2), "parity": p2,
                                   "meets_1e-3": bool(p2 and p2["meets_1e-3"]),
                                   "strict_backward": {"backward_products": 3, "steps": ks3, "ms_per_step": round(ms3, 3), "posts_per_s": round(B / (ms3 * 1e-3), 1),
                                                       "model_frac_of_peak": frac(ms3), "slowdown_vs_headline": round(ms3 / ms_step, 2),
                                                       "gradients": "within 1e-3 relative L2 per tensor (measured 4e-5); same forward, same parity block"},
                                   "note": "same workload, same full train step, same run; the dtype whose per-post outputs meet north_star's 1e-3 against the reference's golden vectors"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
