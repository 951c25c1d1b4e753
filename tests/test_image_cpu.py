"""Image leg of the input pipeline, CPU side: the oracle (oracle/image_oracle.py) against the installed Pillow and the
golden vectors from the transformers ViT image processor; the C library's host-side resampling plan against the oracle."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import image_oracle as IO  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_golden.npz")
SIZES = [(224, 224), (37, 53), (500, 375), (375, 500), (1, 1), (2, 900), (900, 2), (224, 500), (500, 224), (225, 223), (300, 3), (301, 3),
         (3000, 17), (448, 448), (1, 224), (224, 1)]


def pil_order_resize(img, oh, ow):
    """oracle with the installed Pillow's pass order: vertical pass first when h > 100 w (observed in 12.2.0)"""
    h, w, _ = img.shape
    if h > 100 * w:
        return IO.resize_bilinear_u8(np.ascontiguousarray(img.transpose(1, 0, 2)), ow, oh).transpose(1, 0, 2)
    return IO.resize_bilinear_u8(img, oh, ow)


@pytest.mark.parametrize("h,w", SIZES)
def test_oracle_resize_equals_pillow(h, w):
    from PIL import Image
    rng = np.random.default_rng(h * 7919 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for oh, ow in ((224, 224), (32, 48)):
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(pil_order_resize(img, oh, ow), ref), (h, w, oh, ow)


def test_oracle_matches_hf_processor_golden():
    z = np.load(GOLD)
    lut = IO.normalize_lut()
    for i in range(int(z["n"])):
        img = z[f"img{i}"]
        r224 = pil_order_resize(img, 224, 224)
        assert hashlib.sha256(r224.tobytes()).digest() == z[f"u8_224_sha_{i}"].tobytes(), i
        pv = np.ascontiguousarray(np.stack([lut[c][r224[:, :, c]] for c in range(3)]))
        assert hashlib.sha256(pv.tobytes()).digest() == z[f"sha224_{i}"].tobytes(), i
        r32 = pil_order_resize(img, 32, 32)
        assert np.array_equal(np.stack([lut[c][r32[:, :, c]] for c in range(3)]), z[f"pv32_{i}"]), i


def test_plan_from_c_library_equals_oracle():
    """mmhip_image_plan_build is host-only: windows and 22-bit weights must equal the oracle's (Pillow's) for every axis"""
    from smtc_amd import _lib
    lib = _lib.lib()
    hs = np.asarray([s[0] for s in SIZES], dtype=np.int32)
    ws = np.asarray([s[1] for s in SIZES], dtype=np.int32)
    n, S = len(SIZES), 224
    offs = np.arange(n, dtype=np.uint64) * 4096
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    words = int(lib.mmhip_image_plan_words(n, vp(hs), vp(ws), S))
    assert words > 0
    plan = np.zeros(words, dtype=np.int32)
    assert lib.mmhip_image_plan_build(n, vp(offs), vp(hs), vp(ws), S, vp(plan), words) == 0
    assert plan[1] == n and plan[2] == S and plan[3] == words
    for i, (h, w) in enumerate(SIZES):
        r = plan[8 + 16 * i: 8 + 16 * (i + 1)]
        assert (r[2], r[3]) == (h, w) and r[0] == 4096 * i and r[4] == (1 if h > 100 * w else 0)
        for size, k_w, b_off, k_off in ((w, r[5], r[11], r[12]), (h, r[6], r[13], r[14])):
            ksize, bounds, coeffs = IO.precompute_coeffs(size, S)
            assert ksize == k_w
            assert np.array_equal(plan[b_off: b_off + 2 * S].reshape(S, 2), bounds)
            assert np.array_equal(plan[k_off: k_off + S * ksize].reshape(S, ksize), coeffs)
        if r[4] == 0:
            _, bv, _ = IO.precompute_coeffs(h, S)
            assert r[7] == bv[0, 0] and r[8] == bv[-1, 0] + bv[-1, 1] - bv[0, 0]
    assert lib.mmhip_image_plan_tmp_bytes(vp(plan)) > 0
    # errors: zero-sized image, too small a buffer
    bad = np.asarray([0], dtype=np.int32)
    assert lib.mmhip_image_plan_words(1, vp(bad), vp(bad), S) == 0
    assert lib.mmhip_image_plan_build(n, vp(offs), vp(hs), vp(ws), S, vp(plan), words - 1) == -3
