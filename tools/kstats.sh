#!/bin/bash
# usage (GPU box): tools/kstats.sh NAME [bench args...]  -> gpurun_out/NAME_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py)
NAME=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/ks_$NAME -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-parity "$@" > /tmp/ks_$NAME.log 2>&1 || { tail -5 /tmp/ks_$NAME.log; exit 1; }
cp $(find /tmp/ks_$NAME -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${NAME}_kernel_stats.csv
