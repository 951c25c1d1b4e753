"""Throughput of the GPU image processor (csrc/image.hip) on a batch of photo-sized images, inputs resident in HBM, next
to Pillow + numpy on the host.  Prints one JSON line (metric images/s, HBM roofline of the two kernel launches)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd import _lib
from smtc_amd.image_processing import GpuImageProcessor

n, h, w, S = 64, 768, 1024, 224
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]
p = GpuImageProcessor(device="cuda:0")
lib = _lib.lib()
packed, plan, _ = p.pack(imgs)
packed_d, plan_d = packed.cuda(), plan.cuda()
tmp = torch.empty(int(lib.mmhip_image_plan_tmp_bytes(C.c_void_p(plan.data_ptr()))), dtype=torch.uint8, device="cuda")
out = torch.empty(n, 3, S, S, device="cuda")
run = lambda: _lib.check(lib.mmhip_image_preprocess(_lib.ptr(packed_d), C.c_void_p(plan.data_ptr()), _lib.ptr(plan_d), _lib.ptr(p.lut()), _lib.ptr(out), None,
                                                    _lib.ptr(tmp), _lib.stream_ptr()))
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 50
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
t0 = time.perf_counter(); pk = p.pack(imgs); host_pack_ms = (time.perf_counter() - t0) * 1e3
from PIL import Image
x = (np.arange(256).astype(np.float64) * (1 / 255)).astype(np.float32); lut = (x - np.float32(0.5)) / np.float32(0.5)
t0 = time.perf_counter()
for im in imgs[:16]:
    r = np.asarray(Image.fromarray(im).resize((S, S), Image.BILINEAR)); _ = lut[r].transpose(2, 0, 1).copy()
cpu_ms_per_img = (time.perf_counter() - t0) / 16 * 1e3
alg = n * (h * w * 3 + 3 * S * S * 4)
print(json.dumps({"metric": "images/sec (resize 224 PIL-bilinear + normalize), 1024x768 RGB", "value": round(n / (ms * 1e-3), 1), "unit": "images/s",
                  "ms_per_batch": round(ms, 4), "batch": n, "dtype": "u8", "roofline": {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                  "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / 8e12, 4), "algorithmic_bytes_per_batch": alg},
                  "host_pack_plan_ms_per_batch": round(host_pack_ms, 2),
                  "cpu_baseline": {"value": round(1e3 / cpu_ms_per_img, 1), "unit": "images/s", "cores": 1, "kind": "reference", "sample": "16 images, PIL resize + numpy LUT"}}))
