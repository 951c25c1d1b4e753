#!/usr/bin/env python3
"""Which single-pass numerics policy meets north_star's 1e-3 on the per-post outputs?  (VERDICT r2 #3b)

CPU only.  Runs the reference's four forward goldens (tests/golden/fwd_*.npz: XLM-R, BERT, concat, full depth) through the
oracle under a rounding policy (oracle/mm_oracle.py `rounding`): which tensors are rounded to which 16-bit type where the HIP
path would store them or feed them to an MFMA.  Prints max|got - ref| / max|ref| per output, worst over the goldens.

  python tools/numerics_study.py [--out profiles/r03_numerics_study.txt]

Policies (op_a / op_w = activation / weight operand of every tower matrix product; st_act = stored qkv, ctx, FC1 output;
st_resid = stored LayerNorm inputs and the image tower's residual stream; st_ln = stored LayerNorm outputs):
"""
import argparse
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mm_oracle as O  # noqa: E402

GOLDENS = ["fwd_small_xlmr", "fwd_small_bert", "fwd_small_concat", "fwd_full_xlmr",
           # round 4: nine more reference-generated cases (tests/golden/make_golden.py --extra): seeds, batch sizes, lengths, padding, depth
           "fwd_x_full_xlmr_a", "fwd_x_full_xlmr_b", "fwd_x_full_xlmr_c", "fwd_x_full_bert_a", "fwd_x_full_bert_b", "fwd_x_full_concat",
           "fwd_x_mid_xlmr", "fwd_x_small_xlmr", "fwd_x_small_bert"]
KEYS = ["out_cls", "logits_per_text", "out_tim", "mm_features"]

POLICIES = [
    # name, kwargs of O.rounding, MFMA passes per product (cost), bytes per stored activation element (LN-in / LN-out / other)
    ("bf16 (shipped throughput mode)", dict(round_operands="bf16"), 1),
    ("f16 (shipped)", dict(round_operands="f16"), 1),
    ("f16 operands, fp32 residual stream (LN inputs stored fp32)", dict(round_operands="f16", st_resid=None), 1),
    ("f16 operands, fp32 LN inputs AND outputs (rounded to f16 only as MFMA operand)", dict(round_operands="f16", st_resid=None, st_ln=None), 1),
    ("f16 operands, every stored activation fp32 (operand rounding only)", dict(round_operands=None, op_a="f16", op_w="f16"), 1),
    ("bf16, fp32 residual stream", dict(round_operands="bf16", st_resid=None), 1),
    ("bf16, hi+lo activations at the LN outputs only (A operand 2 passes where A is an LN output)", dict(round_operands="bf16", st_ln="bf16x2", st_resid=None), 1.5),
    ("bf16 weights, hi+lo (16-bit-mantissa) activations everywhere: 2 passes", dict(round_operands=None, op_a="bf16x2", op_w="bf16", st_act="bf16x2", st_resid=None, st_ln="bf16x2"), 2),
    ("f16 activations (stored f16), f16 hi+lo weights: 2 passes", dict(round_operands="f16", op_w="f16x2"), 2),
    ("f16 operand rounding of fp32-stored activations, f16 hi+lo weights: 2 passes", dict(round_operands=None, op_a="f16", op_w="f16x2"), 2),
    ("f16 hi+lo activations (stored fp32), f16 weights: 2 passes", dict(round_operands=None, op_a="f16x2", op_w="f16"), 2),
    ("f16 activations stored f16 except fp32 LN inputs/outputs, f16 hi+lo weights: 2 passes", dict(round_operands="f16", op_w="f16x2", st_resid=None, st_ln=None), 2),
    ("bf16x3-like: hi+lo activations AND weights (3 passes), fp32 stores", dict(round_operands=None, op_a="bf16x2", op_w="bf16x2"), 3),
    # round 4: activations STORED as 16-bit (hi, lo) pairs -- the planes the matrix cores read, written by the producers' epilogues
    ("f16 hi+lo activations stored as f16 pairs (22 bits), f16 weights: 2 passes", dict(round_operands=None, op_a="f16x2", op_w="f16", st_act="f16x2", st_resid="f16x2", st_ln="f16x2"), 2),
    ("bf16 hi+lo activations stored as bf16 pairs (16 bits), bf16 hi+lo weights: 3 passes", dict(round_operands=None, op_a="bf16x2", op_w="bf16x2", st_act="bf16x2", st_resid="bf16x2", st_ln="bf16x2"), 3),
    ("bf16 pairs for GEMM-feeding tensors only (fp32 residual stream), bf16 hi+lo weights: 3 passes", dict(round_operands=None, op_a="bf16x2", op_w="bf16x2", st_act="bf16x2", st_ln="bf16x2"), 3),
    ("f16 hi+lo activations stored as f16 pairs, f16 hi+lo weights: 3 passes", dict(round_operands=None, op_a="f16x2", op_w="f16x2", st_act="f16x2", st_resid="f16x2", st_ln="f16x2"), 3),
]


def load(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    cfg = O.OracleConfig(**ast.literal_eval(str(z["cfg"])))
    return z, cfg


def run(policy_kwargs):
    worst = {k: 0.0 for k in KEYS}
    run.per_golden = {}
    for name in GOLDENS:
        z, cfg = load(name)
        P = O.make_params(cfg, int(z["seed_w"]))
        ids, mask, pixels, _ = O.synthetic_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed_x"]), bool(z["pad"]))
        tim = (torch.from_numpy(z["tim_ids"]), torch.from_numpy(z["tim_mask"]))
        with torch.no_grad(), O.rounding(**policy_kwargs):
            out_cls, lpt, out_tim, _, feats = O.mm_forward(P, ids, mask, pixels, cfg, tim)
        for got, key in ((out_cls, "out_cls"), (lpt, "logits_per_text"), (out_tim, "out_tim"), (feats, "mm_features")):
            ref = torch.from_numpy(z[key])
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            worst[key] = max(worst[key], err)
            run.per_golden[name] = max(run.per_golden.get(name, 0.0), err)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--per-golden", action="store_true", help="also print the worst output error of every golden")
    ap.add_argument("--only", default="", help="substring filter on the policy name")
    args = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    lines = ["# numerics study: max|got - ref| / max|ref| against the reference's fp32 golden vectors, worst of %d forward goldens (CPU emulation)" % len(GOLDENS),
             "# north_star tolerance: 1e-3 on every per-post output",
             "%-96s %6s %10s %10s %10s %10s  %s" % ("policy", "passes", *KEYS, "meets 1e-3")]
    for name, kw, passes in POLICIES:
        if args.only and args.only not in name:
            continue
        w = run(kw)
        ok = all(v < 1e-3 for v in w.values())
        lines.append("%-96s %6s %10.2e %10.2e %10.2e %10.2e  %s" % (name, passes, *(w[k] for k in KEYS), "YES" if ok else "no"))
        print(lines[-1], flush=True)
        if args.per_golden:
            lines.append("    per golden (worst output): " + "  ".join("%s %.1e" % (g.replace("fwd_", ""), e) for g, e in run.per_golden.items()))
            print(lines[-1], flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
