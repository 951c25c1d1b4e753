"""-m gpu: the run_mm_late.py mirror end to end on synthetic posts (files, CSV layout, checkpoint round trip), and the
data-parallel step with two ranks on one card (gloo moves the device tensors; RCCL needs one GPU per rank)."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, **kw):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(kw.pop("env", {}))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def test_cli_train_save_load(tmp_path):
    res = str(tmp_path) + "/"
    base = [sys.executable, "-m", "smtc_amd.run_mm_late", "--txt_model_name", "bernice", "--img_model_name", "vit", "--fusion_name", "attention",
            "--task", "3", "--use_clip_loss", "--use_tim_loss", "--synthetic", "--n_synthetic", "32", "--batch_size", "8", "--epochs", "1",
            "--arch_layers", "1", "--results_dir", res, "--save_model", "--save_preds"]
    run(base)
    stem = res + "bernice-vit-attention_task3_seed30_itc0.1itm0.1_"
    for suffix in ("net.pth", "metrics_val.csv", "metrics_test.csv", "preds.csv"):
        assert os.path.exists(stem + suffix), suffix
    mv = pd.read_csv(stem + "metrics_val.csv")
    assert list(mv.columns) == ["metric", "epoch-1"]
    assert list(mv.metric) == ["f1_weighted", "f1_macro", "precision_weighted", "precision_macro", "recall_weighted", "recall_macro", "loss"]
    assert np.isfinite(mv["epoch-1"]).all()
    assert list(pd.read_csv(stem + "preds.csv").columns) == ["data_id", "label", "prediction"]
    sd = torch.load(stem + "net.pth", map_location="cpu")
    assert "dual_encoder.text_model.encoder.layer.0.attention.self.query.weight" in sd and "linear_fusion.weight" in sd
    assert "dual_encoder.vision_model.encoder.layer.0.attention.attention.query.weight" in sd
    run([a for a in base if a not in ("--save_model", "--save_preds")] + ["--load_saved_model"])
    assert os.path.exists(stem + "preds_lm.csv") and os.path.exists(stem + "metrics_lm.csv")


DP_SCRIPT = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.environ["ROOT"])
import smtc_amd
from smtc_amd import dist as mmdist
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch
os.environ["LOCAL_RANK"] = "0"                      # both ranks share the one card
mmdist.init_from_env(backend="gloo")
rank, world = mmdist.rank(), mmdist.world_size()
cfg = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=64, dropout=0.0)
arch = dict(layers_txt=2, layers_img=1, vocab=300, max_pos=130, p_hidden=0.0, p_attn=0.0)
tr = MMLate_Model(cfg, "bernice", "vit", "attention", arch=arch, seed=5)
B = 4
ids, mask, px, oh = synthetic_batch(300, 3, B * world, 64, 77, pad=True)
sl = slice(rank * B, (rank + 1) * B)
np.random.seed(30 + rank)
tim = tr.prepare_itm_inputs(ids[sl].cuda(), mask[sl].cuda())
loss, _ = tr.train_step(ids[sl].cuda(), mask[sl].cuda(), px[sl], oh[sl], None, 1e-3, 0.00025, 1, tim=tim)
torch.save({"p": tr.model._flat_train.cpu(), "tim": [t.cpu() for t in tim], "loss": loss.cpu()}, os.environ["OUT"] + f"/rank{rank}.pt")
torch.distributed.barrier()
if rank == 0:
    # single-process reference: gradients of the two rank-local losses averaged, one AdamW step
    ref = MMLate_Model(cfg, "bernice", "vit", "attention", arch=arch, seed=5)
    assert torch.equal(ref.model._flat_train.cpu(), torch.load(os.environ["OUT"] + "/init.pt")) if os.path.exists(os.environ["OUT"] + "/init.pt") else True
    from smtc_amd import _lib
    import ctypes as C
    lib, m = _lib.lib(), ref.model
    g = torch.zeros_like(m._flat_grad)
    for r in range(world):
        s2 = slice(r * B, (r + 1) * B)
        t = [x.cuda() for x in torch.load(os.environ["OUT"] + f"/rank{r}.pt")["tim"]]
        m.train(); m._flat_grad.zero_()
        m._engine_forward(ids[s2].cuda(), mask[s2].cuda(), px[s2], t[0], t[1])
        lo = torch.empty(4, device="cuda"); ohd = oh[s2].cuda().contiguous()
        _lib.check(lib.mmhip_loss(m._handle, _lib.ptr(ohd), None, _lib.ptr(t[2]), 0.8, 0.1, 0.1, _lib.ptr(lo), None, _lib.stream_ptr()))
        _lib.check(lib.mmhip_backward(m._handle, None, None, None, None, _lib.stream_ptr()))
        g += m._flat_grad
    m._flat_grad.copy_(g)
    ref.world = world
    ref._adamw(1e-3, 0.00025, 1)
    got = torch.load(os.environ["OUT"] + "/rank0.pt")["p"]
    other = torch.load(os.environ["OUT"] + "/rank1.pt")["p"]
    err = (got - m._flat_train.cpu()).abs().max().item()
    print("DP_ERR", err, "RANKS_EQUAL", torch.equal(got, other))
torch.distributed.destroy_process_group()
'''


@pytest.mark.parametrize("bucket_mb", ["48", "8"])
@pytest.mark.parametrize("opt", ["allreduce", "shard"])
def test_two_rank_step_equals_averaged_single_process(tmp_path, opt, bucket_mb):
    """opt = shard: reduce-scatter -> AdamW on the rank's own shard -> all-gather of the parameters (dist.ShardedBuckets, MMHIP_DP_OPT=shard)
    instead of all-reduce + replicated AdamW -- the same parameters after the step, replicas bit-identical.
    bucket_mb = 8: every text layer (28 MB of gradients) leaves as its own collective, so the PER-BUCKET optimizer runs (include/mmhip.h
    MMHIP_CB_BUCKET: the layer's AdamW + refresh on the engine's side stream behind its own collective, beside the stages below); 48: one bucket,
    the single barrier of round 4"""
    script = tmp_path / "dp.py"
    script.write_text(DP_SCRIPT)
    r = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
             str(29600 + os.getpid() % 300 + (17 if opt == "shard" else 0) + (31 if bucket_mb == "8" else 0)), str(script)],
            env={"ROOT": ROOT, "OUT": str(tmp_path), "MMHIP_DP_OPT": opt, "MMHIP_BUCKET_MB": bucket_mb})
    line = [l for l in r.stdout.splitlines() if l.startswith("DP_ERR")][0].split()
    assert line[3] == "True", line                      # replicas stay bit-identical
    assert float(line[1]) < 2e-6, line                  # == one process on the averaged gradients (fp32 sum order only)


def test_cli_config0_real_pipeline(tmp_path):
    """BASELINE configs[0]: run_mm_late.py --txt_model_name bernice --img_model_name vit --fusion_name attention --task 2
    --testing, through the real input pipeline (data key CSV -> tweet normalisation -> tokenizer -> JPEG decode / resize /
    normalise) on a generated dummy task (the reference ships header-only data keys); cwd and relative paths as the
    reference uses them (../data, ../results, ../../../BERNICE)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_dummy_task
    sys.path.insert(0, ROOT)
    import smtc_amd  # noqa: F401
    run_dir = make_dummy_task.main(str(tmp_path), 64, 1)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "smtc_amd.run_mm_late", "--txt_model_name", "bernice", "--img_model_name", "vit", "--fusion_name",
                        "attention", "--task", "2", "--testing", "--use_clip_loss", "--epochs", "1", "--save_preds"],
                       cwd=run_dir, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = os.path.join(os.path.dirname(run_dir), "results", "mm_late", "testing")
    stem = os.path.join(out, "bernice-vit-attention_task2_seed30_itc0.1_")
    mv = pd.read_csv(stem + "metrics_val.csv")
    assert list(mv.columns) == ["metric", "epoch-1"] and np.isfinite(mv["epoch-1"]).all()
    preds = pd.read_csv(stem + "preds.csv")
    assert set(preds.prediction.unique()) <= {0, 1, 2, 3} and preds.data_id.between(1000, 1063).all()
    # three epochs with both auxiliary losses and the image-tower output cache: epochs 2 and 3 take every post from the cache
    r = subprocess.run([sys.executable, "-m", "smtc_amd.run_mm_late", "--txt_model_name", "bernice", "--img_model_name", "vit", "--fusion_name",
                        "attention", "--task", "2", "--testing", "--use_clip_loss", "--use_tim_loss", "--epochs", "3", "--cache_vision", "64",
                        "--seed", "31", "--num_workers", "2"], cwd=run_dir, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    mv = pd.read_csv(os.path.join(out, "bernice-vit-attention_task2_seed31_itc0.1itm0.1_metrics_val.csv"))
    assert list(mv.columns) == ["metric", "epoch-1", "epoch-2", "epoch-3"] and np.isfinite(mv.values[:, 1:].astype(float)).all()
    import re
    m = re.search(r"image-tower output cache: (\d+) posts served from HBM, (\d+) computed, (\d+) cached", r.stdout + r.stderr)
    assert m and int(m.group(3)) == 64 and int(m.group(2)) == 64 and int(m.group(1)) == 2 * 64, m and m.groups()


def test_epoch_prefetch_twin_loader_trains_every_epoch(tmp_path):
    """MMHIP_EPOCH_PREFETCH=1 (round 5): epochs alternate between the training loader and its twin, the next epoch's loader is primed while this
    epoch's tail is trained on.  Three epochs through the real pipeline with worker processes and the image ring: the run completes, every epoch
    has finite validation metrics, and the ring ends with every slot free (nothing leaked across the primed iterators).  The default (flag off) is
    the reference's one iteration per epoch and keeps its seeded order -- covered by test_cli_config0_real_pipeline."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_dummy_task
    import smtc_amd  # noqa: F401
    run_dir = make_dummy_task.main(str(tmp_path), 96, 1)
    out = os.path.join(os.path.dirname(run_dir), "results", "mm_late", "testing")
    env = dict(os.environ, PYTHONPATH=ROOT, MMHIP_EPOCH_PREFETCH="1")
    r = subprocess.run([sys.executable, "-m", "smtc_amd.run_mm_late", "--txt_model_name", "bernice", "--img_model_name", "vit", "--fusion_name", "attention",
                        "--task", "2", "--testing", "--use_clip_loss", "--use_tim_loss", "--epochs", "3", "--batch_size", "16", "--seed", "33",
                        "--num_workers", "2", "--save_preds"], cwd=run_dir, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    mv = pd.read_csv(os.path.join(out, "bernice-vit-attention_task2_seed33_itc0.1itm0.1_metrics_val.csv"))
    assert list(mv.columns) == ["metric", "epoch-1", "epoch-2", "epoch-3"] and np.isfinite(mv.values[:, 1:].astype(float)).all()
    assert r.stdout.count("Epoch:") == 3


RCCL_SCRIPT = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.environ["ROOT"])
import smtc_amd
from smtc_amd import dist as mmdist
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:" + os.environ["PORT"], rank=0, world_size=1)
cfg = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=64, dropout=0.05)
arch = dict(layers_txt=2, layers_img=1, vocab=300, max_pos=130)
ids, mask, px, oh = synthetic_batch(300, 3, 4, 64, 77, pad=True)
os.environ["MMHIP_DETERMINISTIC"] = "1"    # single-writer reductions: identical arithmetic gives identical bits, so three steps compare exactly
out = {}
# exchange forced through RCCL: the native data-parallel step (mmhip_train_step_dp + callbacks) and the staged Python step; no exchange: the native step
for name, force, native_dp, opt in (("rccl_native", "1", "1", "allreduce"), ("rccl_staged", "1", "0", "allreduce"), ("rccl_shard", "1", "1", "shard"),
                                    ("plain", "0", "1", "allreduce")):
    os.environ["MMHIP_FORCE_EXCHANGE"] = force
    os.environ["MMHIP_NATIVE_DP"] = native_dp
    os.environ["MMHIP_DP_OPT"] = opt          # shard: in-place reduce_scatter_tensor -> AdamW on the rank's shard (all of it at one rank) -> in-place all_gather_into_tensor
    tr = MMLate_Model(cfg, "bernice", "vit", "attention", arch=arch, seed=5)
    np.random.seed(30)
    losses = []
    for step in (1, 2, 3):
        loss, _ = tr.train_step(ids.cuda(), mask.cuda(), px, oh, None, 1e-3, 0.00025, step)
        losses.append(float(loss[0]))
    torch.cuda.synchronize()
    mom = tr._opt[0].clone() if opt == "allreduce" else None          # (the sharded optimizer keeps its dense moments per shard, elsewhere)
    out[name] = (tr.model._flat_train.clone(), mom, tr.model._word_row_state.clone(), losses, int(tr.model._last.get("exchange_bytes", 0)))
ref = out["plain"]
ok = {k: all(a is None or torch.equal(a, b) for a, b in zip(v[:3], ref[:3])) and v[3] == ref[3] for k, v in out.items()}
moved = float((ref[0] - MMLate_Model(cfg, "bernice", "vit", "attention", arch=arch, seed=5).model._flat_train).abs().max())
print("RCCL_EQ", ok["rccl_native"], ok["rccl_staged"], torch.distributed.get_backend(), out["rccl_native"][4] > 0, out["rccl_staged"][4] > 0, moved > 1e-4, ok["rccl_shard"], ref[3])
torch.distributed.destroy_process_group()
'''


@pytest.mark.parametrize("bucket_mb", ["48", "8"])
def test_rccl_call_pattern_at_world_size_one(tmp_path, bucket_mb):
    """the all-reduce / row-sparse all_gather exchange issued through RCCL itself (backend "nccl", one rank, exchange forced), once by the
    native data-parallel step (mmhip_train_step_dp: the library enqueues, this process starts / finishes the collectives from its callbacks)
    and once by the staged Python step: after three steps with ITC + ITM and dropout the parameters, moments, row flags and the loss
    trajectory are BIT-IDENTICAL to the run without any collective (MMHIP_DETERMINISTIC=1: single-writer reductions).  The one-GPU box
    cannot host two RCCL ranks, so this pins the backend's call pattern (slices of the flat gradient, async work handles, int64 / fp32
    all_gather); tests/test_dist_cpu.py and the two-rank gloo tests pin the arithmetic across ranks."""
    script = tmp_path / "rccl.py"
    script.write_text(RCCL_SCRIPT)
    # bucket_mb = 8: every text layer leaves as its own collective and the per-bucket optimizer runs (MMHIP_CB_BUCKET / MMHIP_CB_WAIT_BUCKET: the
    # engine's side stream waits for the RCCL work stream-side, the layer's AdamW + refresh follow beside the stages below) -- same bits
    r = run([sys.executable, str(script)], env={"ROOT": ROOT, "PORT": str(29900 + os.getpid() % 90 + (5 if bucket_mb == "8" else 0)), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                                                "MMHIP_BUCKET_MB": bucket_mb})
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_EQ")][0].split()
    assert line[1] == "True" and line[2] == "True" and line[3] == "nccl" and line[4] == "True" and line[5] == "True" and line[6] == "True", line
    assert line[7] == "True", line          # round 4: the sharded optimizer's in-place reduce-scatter / all-gather through RCCL, same bits


def test_cli_two_ranks_shard_the_real_data(tmp_path):
    """the CLI under data parallelism on real data keys: two ranks (gloo, sharing the one card) each train on their own shard of
    the dummy task (DistributedSampler), exchange gradients, and end with identical parameters; rank 0 writes the files"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_dummy_task
    run_dir = make_dummy_task.main(str(tmp_path), 64, 1)
    port = str(29700 + os.getpid() % 200)
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, PYTHONPATH=ROOT, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   MMHIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "smtc_amd.run_mm_late", "--txt_model_name", "bernice", "--img_model_name", "vit",
                                       "--fusion_name", "attention", "--task", "2", "--testing", "--use_clip_loss", "--use_tim_loss", "--epochs", "2",
                                       "--batch_size", "8", "--save_model", "--results_dir", str(tmp_path) + f"/res{rank}/"],
                                      cwd=run_dir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-3000:]
    stem = "testing/bernice-vit-attention_task2_seed30_itc0.1itm0.1_"
    assert os.path.exists(str(tmp_path) + "/res0/" + stem + "net.pth") and not os.path.exists(str(tmp_path) + "/res1/" + stem + "net.pth")
    mv = pd.read_csv(str(tmp_path) + "/res0/" + stem + "metrics_val.csv")
    assert list(mv.columns) == ["metric", "epoch-1", "epoch-2"] and np.isfinite(mv.values[:, 1:].astype(float)).all()
