#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per-queue busy time, union busy time, gaps with no
kernel on the chip, and the phases (forward = up to the first backward kernel, backward, optimizer).
usage: [TIMELINE_WINDOW=a,b] timeline.py <dir-or-csv> [step_index_from_end]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    keys = rows[0].keys()
    name_k = "Kernel_Name" if "Kernel_Name" in keys else [k for k in keys if "ame" in k][0]
    q_k = "Queue_Id" if "Queue_Id" in keys else None
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k], r[q_k] if q_k else "0") for r in rows))
    # steps are delimited by the word-table AdamW kernel (one per step, the last optimizer kernel on the main queue)
    # (TIMELINE_MARK: another once-per-step kernel, e.g. maxpool_bwd for the early-fusion engine whose step has no word-row AdamW)
    mark = os.environ.get("TIMELINE_MARK", "adamw_rows_kernel")
    marks = [i for i, e in enumerate(ev) if mark in e[2]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    a, b = marks[-which - 1], marks[-which]
    seg = ev[a + 1: b + 1]
    # weight-refresh kernels that follow this step's last AdamW belong to it; the previous step's do not
    j = b + 1
    while j < len(ev) and "cast" in ev[j][2]:
        seg.append(ev[j]); j += 1
    while seg and "cast" in seg[0][2]:
        seg.pop(0)
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    print("step span %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
    if os.environ.get("TIMELINE_WINDOW"):          # "a,b" in ms from the step's start: every kernel that overlaps the window (start, duration, queue, name)
        wa, wb = (float(x) * 1e6 for x in os.environ["TIMELINE_WINDOW"].split(","))
        for s_, e_, n_, q_ in sorted(seg):
            if e_ - t0 >= wa and s_ - t0 <= wb:
                print("    %8.3f ms  %7.1f us  q%s  %s" % ((s_ - t0) / 1e6, (e_ - s_) / 1e3, q_, n_[:110]))
    byq = {}
    for s, e, n, q in seg:
        byq.setdefault(q, []).append((s, e, n))
    for q, v in byq.items():
        busy = sum(e - s for s, e, _ in v)
        print("  queue %s: %4d kernels, busy %.3f ms, from %.3f to %.3f ms" % (q, len(v), busy / 1e6, (v[0][0] - t0) / 1e6, (max(x[1] for x in v) - t0) / 1e6))
    # union busy / idle gaps
    iv = sorted((s, e) for s, e, _, _ in seg)
    cur_s, cur_e = iv[0]
    union, gaps = 0, []
    for s, e in iv[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            gaps.append((cur_e, s))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print("  union busy %.3f ms, idle %.3f ms in %d gaps (largest %.1f us)" % (union / 1e6, sum(g[1] - g[0] for g in gaps) / 1e6, len(gaps), max((g[1] - g[0] for g in gaps), default=0) / 1e3))
    # overlap: time with >= 2 kernels resident
    pts = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    depth, last, multi = 0, pts[0][0], 0
    for t, d in pts:
        if depth >= 2:
            multi += t - last
        depth += d; last = t
    print("  time with >= 2 kernels in flight %.3f ms" % (multi / 1e6))
    # phases on the main queue (the one with most kernels)
    mq = max(byq, key=lambda q: len(byq[q]))
    first_bwd = next((s for s, e, n in byq[mq] if "bwd" in n or "loss" in n), None)
    first_opt = next((s for s, e, n in byq[mq] if "adamw" in n), None)      # first optimizer kernel on the main queue (heads / embeddings)
    if first_bwd and first_opt:
        print("  forward %.3f ms | backward %.3f ms | optimizer + refresh %.3f ms" % ((first_bwd - t0) / 1e6, (first_opt - first_bwd) / 1e6, (t1 - first_opt) / 1e6))
    for q, v in byq.items():
        if q != mq:
            print("  side queue %s ends at %.3f ms; kernels: %s" % (q, (max(x[1] for x in v) - t0) / 1e6, ", ".join(sorted({n.split('(')[0][-40:] for _, _, n in v}))[:300]))
    if os.environ.get("DUMP"):
        for s, e, n, q in seg:
            print("%9.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:90]))


if __name__ == "__main__":
    main()
