#!/usr/bin/env python3
"""HBM-bound kernels of the step: algorithmic bytes per launch / average duration from a rocprofv3 kernel-stats CSV of
`MMHIP_OVERLAP=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline` (kernels one at a time).  hbm_table.py <kernel_stats.csv>"""
import csv, sys
B, T, H, I, P, V = 64, 128, 768, 3072, 197, 250002
Mt, Mv = B * T, B * P
dense = 283856649 - V * H                                    # trainable parameters outside the word table
alg = {   # kernel-name fragment -> (what, algorithmic bytes per launch)
    "adamw_kernel": ("dense AdamW: p,g,m,v read + p,m,v,g written, fp32", dense * 32),
    "adamw_rows_kernel": ("word table, row-lazy AdamW: only rows with a gradient or moments are touched (<= B*T rows of 32 B/param; rows that only decay by exactly 1.0f are skipped)", min(V, Mt) * H * 32),
    "ln_fwd_kernel": ("LayerNorm fwd: x read, y written (16-bit), text rows", Mt * H * 4),
    "ln_bwd_kernel": ("LayerNorm bwd: dy, x read; dx (+ dropout copy) written", Mt * H * 8),
    "cast_dual_kernel": ("weight refresh of a layer: fp32 read, 16-bit copy + transposed copy written", 7087872 * 8),
    "embed_bwd_kernel": ("embedding backward: dx, xhat read; word-row atomics", Mt * H * 4 + Mt * H * 4),
    "embed_fwd_kernel": ("embedding forward: word/pos rows read (fp32), x + xhat written", Mt * H * 8 + Mt * H * 4),
    "reduce_partials_kernel": ("second stage of column reductions", 2 * 512 * H * 4),
}
rows = list(csv.DictReader(open(sys.argv[1])))
print("| kernel | algorithmic bytes / launch | avg µs | GB/s | of 8 TB/s |\n|---|---|---|---|---|")
steps = next((int(r["Calls"]) for r in rows if "adamw_rows_kernel" in r["Name"]), 1)
for frag, (what, nbytes) in alg.items():
    for r in rows:
        if frag in r["Name"] and not (frag == "adamw_kernel" and "rows" in r["Name"]):
            us = float(r["AverageNs"]) / 1e3
            if frag == "adamw_kernel":      # one launch per text layer + the rest: bytes of a whole step over the launches of a step
                us = float(r["TotalDurationNs"]) / 1e3 / steps
                what += f" ({int(r['Calls']) // steps} launches per step, summed)"
            if frag == "cast_dual_kernel":
                pass
            print(f"| `{frag}` — {what} | {nbytes / 1e6:.1f} MB | {us:.1f} | {nbytes / us / 1e3:.0f} | {nbytes / us / 1e3 / 8000:.0%} |")
            break
