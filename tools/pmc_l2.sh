#!/bin/bash
# L2 hit rate of one GEMM shape: pmc_l2.sh M N K tile [cold]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_l2
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc_l2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > /dev/null 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/pmc_l2/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    if "gemm_nt" not in r["Kernel_Name"]:
        continue
    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
h, m = agg.get("TCC_HIT_sum", [0]), agg.get("TCC_MISS_sum", [0])
n = min(len(h), len(m))
hs, ms = sum(h[-n + 1:]) if n > 1 else sum(h), sum(m[-n + 1:]) if n > 1 else sum(m)
print("shape/tile", sys.argv[1:], "launches", n, "L2 hit rate %.3f" % (hs / max(hs + ms, 1)), "requests/launch %.3e" % ((hs + ms) / max(n - 1, 1)))
PY
