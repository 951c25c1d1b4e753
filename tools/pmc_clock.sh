#!/bin/bash
# Clock the chip holds inside one long NT GEMM and its MFMA-busy share: pmc_clock.sh M N K tile [reps]
# effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X_MICROARCH.md 'DVFS give-back': within 3 % of the in-kernel clock on
# dispatches well over 0.3 ms; reads high on short ones), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI_ACTIVE / 8)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_clk
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_clk -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > /dev/null 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys
rows, trace = [], []
for f in glob.glob("/tmp/pmc_clk/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for f in glob.glob("/tmp/pmc_clk/**/*kernel_trace.csv", recursive=True):
    trace += list(csv.DictReader(open(f)))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace if "gemm_nt" in r["Kernel_Name"]]
act = [float(r["Counter_Value"]) for r in rows if "gemm_nt" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
busy = [float(r["Counter_Value"]) for r in rows if "gemm_nt" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
n = min(len(dur), len(act), len(busy))
if n < 2:
    print("no data", sys.argv[1:]); sys.exit(0)
d, a, b = dur[1:n], act[1:n], busy[1:n]          # skip the first (cold) launch
us = sum(d) / len(d)
clk = sum(a) / len(a) / 8 / us / 1e3              # GHz
M, N, K = (int(x) for x in sys.argv[1:4])
print("shape/tile", sys.argv[1:5], "launches", len(d), "avg %.1f us (profiled)  %.0f TFLOP/s  effective clock %.2f GHz  MFMA busy %.3f  -> MFMA-bound rate at this clock %.0f TFLOP/s"
      % (us, 2.0 * M * N * K / us / 1e6, clk, sum(b) / (1024 * sum(a) / 8), 2500.0 * clk / 2.4))
PY
