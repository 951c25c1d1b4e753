#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace CSV as a table: start offset (us), duration (us), queue, grid, kernel.
usage: step_trace.py <dir-or-csv> [step_index_from_end] > step.txt"""
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("mmhip::", "")
    m = re.match(r"_ZN5mmhip\d+([a-z_0-9]+?)I(.*)", n)
    if m:
        args = re.findall(r"Li(\d+)E|Lb(\d)E", m.group(2))
        n = m.group(1) + "<" + ",".join(a or b for a, b in args) + ">"
    return n[:60]


def main():
    path = sys.argv[1]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Grid_Size_X", r.get("Grid_Size", "0")),
                  r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1"))) for r in rows))
    marks = [i for i, e in enumerate(ev) if "adamw_rows_kernel" in e[2]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    a, b = marks[-which - 1], marks[-which]
    seg = ev[a + 1: b + 1]
    t0 = seg[0][0]
    qs = sorted({e[3] for e in seg})
    print("# step of %d kernels, span %.3f ms; queues %s" % (len(seg), (max(e[1] for e in seg) - t0) / 1e6, qs))
    for s, e, n, q, g, w in seg:
        try:
            wgs = int(g) // max(1, int(w))
        except ValueError:
            wgs = 0
        print("%9.1f %8.1f  q%-2d wg%-6d %s" % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), wgs, short(n)))


if __name__ == "__main__":
    main()
