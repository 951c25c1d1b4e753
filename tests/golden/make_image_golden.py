"""Golden vectors of the image leg, generated in the build container from the installed third-party
implementations the reference calls: transformers' PIL-backed ViT image processor (PIL Image.resize BILINEAR + rescale +
normalize).  Fixtures are data only: input images (synthetic), full outputs at size 32, SHA-256 digests at size 224.

    python tests/golden/make_image_golden.py      # writes tests/golden/image_golden.npz
"""
import hashlib
import os

import numpy as np
from PIL import Image
from transformers.models.vit.image_processing_pil_vit import ViTImageProcessorPil

HERE = os.path.dirname(os.path.abspath(__file__))


def synth(h, w, seed):
    """smooth gradients + rectangles + a little noise: compresses well, exercises every filter tap"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 7 % 256)], axis=2).astype(np.int32)
    for _ in range(4):
        y0, x0 = rng.integers(0, h), rng.integers(0, w)
        img[y0: y0 + rng.integers(1, max(h // 2, 2)), x0: x0 + rng.integers(1, max(w // 2, 2))] = rng.integers(0, 256, 3)
    img += rng.integers(-6, 7, img.shape) * (rng.random(img.shape) < 0.1)      # sparse noise keeps the fixture compressible
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    sizes = [(224, 224), (37, 53), (120, 160), (160, 120), (1, 1), (2, 90), (301, 3), (75, 100), (100, 75), (225, 223), (330, 250)]
    out = {"n": np.asarray(len(sizes))}
    p224, p32 = ViTImageProcessorPil(), ViTImageProcessorPil(size={"height": 32, "width": 32})
    for i, (h, w) in enumerate(sizes):
        img = synth(h, w, 100 + i)
        pil = Image.fromarray(img)
        out[f"img{i}"] = img
        a = p224(images=pil, return_tensors="np")["pixel_values"][0]
        b = p32(images=pil, return_tensors="np")["pixel_values"][0]
        assert a.dtype == np.float32 and a.shape == (3, 224, 224)
        out[f"sha224_{i}"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
        out[f"pv32_{i}"] = b
        out[f"u8_224_sha_{i}"] = np.frombuffer(hashlib.sha256(np.asarray(pil.resize((224, 224), Image.BILINEAR)).tobytes()).digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "image_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "image_golden.npz"), os.path.getsize(os.path.join(HERE, "image_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
