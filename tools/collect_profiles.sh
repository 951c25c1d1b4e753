#!/bin/bash
# final evidence of a round, on the GPU box: bench lines, kernel statistics (side streams on / off), PMC passes, diagnostics.
# Everything goes to gpurun_out/final/; copy what is to be judged into profiles/.
set -o pipefail
# usage: collect_profiles.sh a | b     (two gpurun calls: a = PMC passes, bench lines, kernel statistics; b = same-box A/Bs, loader, parity printout)
R=$GRAFT_REPO_ROOT; PART=${1:-a}; O=$R/gpurun_out/final_$PART; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
step() { echo "== $*" >&2; }
if [ "$PART" = a ]; then
step pmc traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-at-tolerance --no-parity > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-at-tolerance --no-parity > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/gemm_traffic.json > /dev/null || exit 1
cp $O/gemm_traffic.json $R/profiles/r05_gemm_traffic.json      # bench.py reports roofline.traffic from the profile of the same kernel sources
step bench
python3 $R/bench.py --steps 20 --warmup 5 --gemm-shapes $O/nt_shapes.txt > $O/bench_cfg2.json 2>$O/bench_cfg2.err || exit 1
python3 $R/bench.py --aux --no-cpu-baseline --no-at-tolerance > $O/bench_cfg3_aux.json 2>/dev/null || exit 1
python3 $R/bench.py --config 4 --no-cpu-baseline > $O/bench_cfg4_224.json 2>/dev/null || exit 1
python3 $R/bench.py --config 4 --image 336 --no-cpu-baseline > $O/bench_cfg4_336.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype f16 --no-cpu-baseline --no-at-tolerance > $O/bench_cfg2_f16.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype bf16x3 --no-cpu-baseline --no-at-tolerance --steps 8 --warmup 3 > $O/bench_cfg2_bf16x3.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype bf16x3 --bwd-products 1 --no-cpu-baseline --no-at-tolerance --steps 8 --warmup 3 --gemm-shapes $O/nt_shapes_bf16x3_bwd1.txt > $O/bench_cfg2_bf16x3_bwd1.json 2>/dev/null || exit 1
python3 $R/bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null || exit 1
python3 $R/bench.py --config 5 --aux --no-cpu-baseline > $O/bench_cfg5_aux.json 2>/dev/null || exit 1
step kernel stats
rocprofv3 --kernel-trace --stats -d /tmp/ks5 -o ks --output-format csv -- python3 $R/bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 || exit 1
cp $(find /tmp/ks5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats.csv
TIMELINE_MARK=maxpool_bwd python3 $R/tools/timeline.py /tmp/ks5 4 > $O/cfg5_timeline.txt
rocprofv3 --kernel-trace --stats -d /tmp/ks_x3 -o ks --output-format csv -- python3 $R/bench.py --dtype bf16x3 --bwd-products 1 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-at-tolerance > /dev/null 2>&1 || exit 1
cp $(find /tmp/ks_x3 -name "*kernel_stats.csv" | head -1) $O/bf16x3_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/ks_on -o ks --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-at-tolerance --no-parity > /dev/null 2>&1 || exit 1
cp $(find /tmp/ks_on -name "*kernel_stats.csv" | head -1) $O/kernel_stats_on.csv
python3 $R/tools/timeline.py /tmp/ks_on 6 > $O/timeline.txt
export MMHIP_OVERLAP=0
rocprofv3 --kernel-trace --stats -d /tmp/ks_off -o ks --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-at-tolerance --no-parity > /dev/null 2>&1 || exit 1
unset MMHIP_OVERLAP
cp $(find /tmp/ks_off -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial.csv
python3 $R/tools/hbm_table.py $O/kernel_stats_serial.csv > $O/hbm_kernels.md
step pmc mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_m -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-at-tolerance --no-parity > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_mfma.py /tmp/pmc_m > $O/mfma_busy.txt
fi
if [ "$PART" = b ]; then
step diagnostics
cd $R
{ BENCH_ARGS="--no-at-tolerance --no-parity" tools/ab_env.sh "BASE=1" "MMHIP_EARLY_ADAMW=0" "MMHIP_OVERLAP=0" "MMHIP_DETERMINISTIC=1"; BENCH_ARGS="--dtype bf16x3 --steps 6 --warmup 3 --no-parity --no-at-tolerance" tools/ab_env.sh "MMHIP_X3_BWD=3" "MMHIP_X3_BWD=2" "MMHIP_X3_BWD=1" "MMHIP_X3_BWD=1 MMHIP_LOCKSTEP=1"; BENCH_ARGS="--config 5 --steps 12 --warmup 4" tools/ab_env.sh "BASE=1" "MMHIP_EARLY_ADAMW=0" "MMHIP_EARLY_STREAMS=0"; } > $O/step_ab.txt 2>/dev/null
python3 tools/loader_bench.py > $O/loader_bench.txt 2>/dev/null
python3 tools/loader_bench.py --epoch_prefetch >> $O/loader_bench.txt 2>/dev/null
python3 -m pytest tests/test_gpu_model.py -q -s -k "train_losses_and_grads or dropout_train_step or forward_matches or config4 or eval_loop" 2>&1 | grep -v "^$" > $O/parity.txt
fi
echo done >&2
