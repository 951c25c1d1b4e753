"""GPU image processor (csrc/image.hip through the C ABI) against Pillow / the oracle / the HF golden vectors: bit-exact."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_golden.npz")


def _proc(size=224):
    from smtc_amd.image_processing import GpuImageProcessor
    return GpuImageProcessor(size=size, device="cuda:0")


def test_batch_of_ragged_images_equals_pillow():
    from PIL import Image
    from oracle import image_oracle as IO
    rng = np.random.default_rng(3)
    sizes = [(224, 224), (37, 53), (500, 375), (375, 500), (1, 1), (2, 900), (900, 2), (224, 500), (500, 224), (225, 223), (300, 3), (301, 3),
             (1500, 11), (448, 448), (1, 224), (224, 1), (768, 1024), (1080, 1920)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    p = _proc()
    packed, plan, n = p.pack(imgs)
    out, u8 = p.run(packed, plan, n, return_u8=True)
    torch.cuda.synchronize()
    lut = IO.normalize_lut()
    for i, img in enumerate(imgs):
        ref = np.asarray(Image.fromarray(img).resize((224, 224), Image.BILINEAR))
        assert np.array_equal(u8[i].cpu().numpy(), ref), sizes[i]
        pv = np.stack([lut[c][ref[:, :, c]] for c in range(3)])
        assert np.array_equal(out[i].cpu().numpy(), pv), sizes[i]


def test_hf_processor_golden():
    z = np.load(GOLD)
    n = int(z["n"])
    imgs = [z[f"img{i}"] for i in range(n)]
    pv = _proc()(imgs)["pixel_values"].cpu().numpy()
    for i in range(n):
        assert hashlib.sha256(np.ascontiguousarray(pv[i]).tobytes()).digest() == z[f"sha224_{i}"].tobytes(), i
    pv32 = _proc(32)(imgs)["pixel_values"].cpu().numpy()
    for i in range(n):
        assert np.array_equal(pv32[i], z[f"pv32_{i}"]), i


def test_pil_inputs_modes_and_errors():
    from PIL import Image
    from smtc_amd import _lib
    rng = np.random.default_rng(5)
    g = Image.fromarray(rng.integers(0, 256, (50, 70), dtype=np.uint8), mode="L")           # grey -> RGB like .convert("RGB")
    rgba = Image.fromarray(rng.integers(0, 256, (40, 30, 4), dtype=np.uint8), mode="RGBA")
    p = _proc()
    pv = p([g, rgba])["pixel_values"]
    from oracle import image_oracle as IO
    for i, im in enumerate((g, rgba)):
        assert np.array_equal(pv[i].cpu().numpy(), IO.preprocess(np.asarray(im.convert("RGB"))))
    with pytest.raises(ValueError):
        p([np.zeros((4, 4), dtype=np.uint8)])
    with pytest.raises(ValueError):
        p([np.zeros((4, 4, 3), dtype=np.float32)])
    assert p([])["pixel_values"].shape == (0, 3, 224, 224)
    # a plan that was never built is rejected before any launch
    bogus = torch.zeros(64, dtype=torch.int32)
    with pytest.raises(_lib.MMHipError):
        p.run(torch.zeros(16, dtype=torch.uint8), bogus, 1)


def test_full_batch_properties_and_prefetcher():
    """B = 64 photo-sized images: (a) equals Pillow per image, (b) order of the batch does not matter, (c) the prefetcher +
    raw-image collate deliver the same tensors as the host form of the pipeline"""
    from PIL import Image
    from smtc_amd.image_processing import DevicePrefetcher, RawImageCollate
    rng = np.random.default_rng(11)
    sizes = [(int(rng.integers(200, 1100)), int(rng.integers(200, 1400))) for _ in range(64)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    p = _proc()
    pv = p(imgs)["pixel_values"]
    perm = rng.permutation(64)
    pv2 = p([imgs[j] for j in perm])["pixel_values"]
    assert torch.equal(pv[perm], pv2)
    x = (np.arange(256).astype(np.float64) * (1 / 255)).astype(np.float32)
    lut = (x - np.float32(0.5)) / np.float32(0.5)
    for i in (0, 17, 63):
        ref = np.asarray(Image.fromarray(imgs[i]).resize((224, 224), Image.BILINEAR))
        assert np.array_equal(pv[i].cpu().numpy(), lut[ref].transpose(2, 0, 1))

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 64

        def __getitem__(self, i):
            return {"image": imgs[i], "input_ids": torch.full((1, 8), i), "labels": torch.tensor([1, 0]), "data_id": torch.tensor(i)}
    loader = torch.utils.data.DataLoader(DS(), batch_size=16, shuffle=False, collate_fn=RawImageCollate(p))
    got = []
    for b in DevicePrefetcher(loader, "cuda:0", p, depth=2):
        assert b["pixel_values"].is_cuda and b["input_ids"].is_cuda and not b["data_id"].is_cuda
        assert int(b["input_ids"][0, 0, 0]) == int(b["data_id"][0])
        got.append(b["pixel_values"].clone())
    assert torch.equal(torch.cat(got), pv)
    # (d) round 4: the same batches through the pinned shared-memory ring, written by worker processes -- bit-identical pixels, every slot
    # back in the free queue after the epoch, and also after an epoch the consumer abandons half-way
    from smtc_amd.image_processing import RingCollate, SharedImageRing
    ring = SharedImageRing(3, 16 * 1100 * 1400 * 3 + 4096, timeout=120.0)      # (a worker waits for a slot as long as it takes: this test is about the bytes)
    try:
        _ring_legs(ring, DS, p, pv)
    finally:
        ring.close()


def _ring_legs(ring, DS, p, pv):
    from smtc_amd.image_processing import DevicePrefetcher, RingCollate
    assert ring.pinned
    # (3 slots: one per batch the loader can have outstanding -- 2 workers x prefetch 1 -- is not enough beside the 2 the prefetcher stages;
    # depth 1 stages one and hands its slot back before it blocks: 2 + 1)
    loader = torch.utils.data.DataLoader(DS(), batch_size=16, shuffle=False, num_workers=2, collate_fn=RingCollate(p, ring), prefetch_factor=1)
    pf = DevicePrefetcher(loader, "cuda:0", p, depth=1, ring=ring)
    got = [b["pixel_values"].clone() for b in pf]
    assert torch.equal(torch.cat(got), pv) and pf.ring_batches == 4 and pf.queue_batches == 0
    free = lambda: sorted(ring.free.get(timeout=2.0) for _ in range(ring.slots))
    slots = free()
    assert slots == [0, 1, 2]
    for s_ in slots:
        ring.release(s_)
    loader = torch.utils.data.DataLoader(DS(), batch_size=16, shuffle=False, num_workers=0, collate_fn=RingCollate(p, ring))
    for i, b in enumerate(DevicePrefetcher(loader, "cuda:0", p, depth=2, ring=ring)):
        assert torch.equal(b["pixel_values"], pv[16 * i: 16 * i + 16])
        if i == 1:
            break                                                   # two more batches are staged: their slots must come back too
    assert free() == [0, 1, 2]
    for s_ in (0, 1, 2):
        ring.release(s_)
    # (e) persistent workers, an epoch the consumer abandons after its first batch: the batches the workers had prefetched for it are dropped by the
    # loader's reset and the slots they name would be gone for good -- the ring's lease book hands them back at the next epoch's
    # start, so the next epoch still travels through the ring, whole
    from smtc_amd.image_processing import SharedImageRing
    with pytest.raises(ValueError):
        DevicePrefetcher([], "cuda:0", p, depth=3, ring=ring)                # 3 slots cannot hold three staged batches and the one being assembled
    ring4 = SharedImageRing(4, 16 * 1100 * 1400 * 3 + 4096, timeout=20.0)
    try:
        loader = torch.utils.data.DataLoader(DS(), batch_size=16, shuffle=False, num_workers=2, collate_fn=RingCollate(p, ring4), prefetch_factor=1,
                                             persistent_workers=True)
        pf = DevicePrefetcher(loader, "cuda:0", p, depth=1, ring=ring4)
        for i, b in enumerate(pf):
            assert torch.equal(b["pixel_values"], pv[:16])
            break
        import time
        time.sleep(1.0)                                                      # (the workers finish what they had been asked for)
        for epoch in range(2):
            n0 = pf.ring_batches
            for i, b in enumerate(pf):
                assert torch.equal(b["pixel_values"], pv[16 * i: 16 * i + 16]) and b["data_id"].tolist() == list(range(16 * i, 16 * i + 16))
            assert i == 3 and pf.ring_batches - n0 == 4 and pf.queue_batches == 0
        del loader, pf
    finally:
        ring4.close()


def test_trimmed_padding_gives_the_same_outputs_and_gradients():
    """DevicePrefetcher.trim: a batch whose longest post has 41 tokens runs with T = 64 instead of 128 -- same logits,
    features, loss and (dropout off) the same gradients"""
    from smtc_amd.image_processing import DevicePrefetcher
    from smtc_amd.mm_late import MM_Model
    from oracle import mm_oracle as O
    arch = dict(layers_txt=2, layers_img=1, vocab=500, max_pos=130, p_hidden=0.0, p_attn=0.0)
    cfg = O.OracleConfig(layers_txt=2, layers_img=1, vocab=500, max_pos=130, num_labels=3)
    model = MM_Model(3, "bernice", "vit", 0.0, "attention", arch=arch, max_posts=8, max_text_len=128, seed=2)
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, 8, 128, 9, True)
    lens = [41, 7, 33, 12, 40, 3, 25, 18]
    for b, n in enumerate(lens):
        mask[b, n:] = 0
        ids[b, n:] = 1                                     # XLM-R pad id
    full = {"input_ids": ids.unsqueeze(1), "attention_mask": mask.unsqueeze(1), "labels": onehot}
    trimmed = DevicePrefetcher.trim(full)
    assert trimmed["input_ids"].shape[-1] == 64 and trimmed["attention_mask"].shape[-1] == 64 and full["input_ids"].shape[-1] == 128
    res = []
    for bt in (full, trimmed):
        model.train()
        model._flat_grad.zero_()
        out = model(bt["input_ids"].squeeze(1), bt["attention_mask"].squeeze(1), pixels)
        loss = O.cls_loss(out[0], onehot.to(out[0].device).float(), None) + 0.1 * out[1].float().pow(2).mean()
        loss.backward()
        res.append((out[0].detach().clone(), out[4].detach().clone(), model._flat_grad.clone()))
    assert torch.allclose(res[0][0], res[1][0], atol=1e-6) and torch.allclose(res[0][1], res[1][1], atol=1e-6)
    rel = (res[0][2] - res[1][2]).norm() / res[0][2].norm()
    assert rel < 1e-5, rel
