"""helpers for the -m gpu tests: device buffers via torch, calls through the C ABI (ctypes)."""
import ctypes as C

import numpy as np
import torch

import smtc_amd  # noqa: F401
from smtc_amd import _lib

DT = {"bf16": (_lib.BF16, torch.bfloat16), "f16": (_lib.F16, torch.float16), "x3": (_lib.F32, torch.float32)}


def dev():
    return torch.device("cuda:0")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    rc = getattr(_lib.lib(), name)(*args)
    assert rc == 0, f"{name} returned {rc}"


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)


def keep_mask(shape, stream_id, seed, p, offset=0):
    """the kernels' dropout keep mask for a row-major tensor of `shape` (oracle/mm_oracle.py hash)"""
    from oracle import mm_oracle as O
    n = int(np.prod(shape))
    return torch.from_numpy(O.hash_keep_mask(n, offset, stream_id, seed, p)).view(*shape), O.keep_scale(p)
