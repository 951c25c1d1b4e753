#!/usr/bin/env python3
"""Uninitialised-read detector for a GPU test: before every repetition the torch caching allocator's free blocks are filled with NaN (or a byte pattern),
so a kernel that reads memory nobody wrote sees poison instead of the zeros of a fresh allocation.  Calls a test FUNCTION of tests/ directly.
    python tools/poison_run.py tests.test_gpu_model:test_backward_product_policy_keeps_outputs_and_bounds_gradients 3 --reps 6 [--gb 24]"""
import argparse
import importlib
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def poison(gb, value):
    blocks = []
    try:
        for _ in range(int(gb * 4)):
            t = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda:0")      # 256 MB
            t.fill_(value)
            blocks.append(t)
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    del blocks          # back to the caching allocator, contents intact


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("target")
    ap.add_argument("args", nargs="*")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--gb", type=float, default=16)
    a = ap.parse_args()
    mod, fn = a.target.split(":")
    f = getattr(importlib.import_module(mod), fn)
    args = [int(x) if x.lstrip("-").isdigit() else x for x in a.args]
    bad = 0
    for r in range(a.reps):
        poison(a.gb, float("nan") if r % 2 == 0 else 3.0e38)
        try:
            f(*args)
            print("rep", r, "ok", flush=True)
        except Exception:
            bad += 1
            print("rep", r, "FAILED", flush=True)
            traceback.print_exc()
    sys.exit(1 if bad else 0)
