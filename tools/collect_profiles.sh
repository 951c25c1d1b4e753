#!/bin/bash
# final evidence of a round, on the GPU box: bench lines, kernel statistics (side streams on / off), PMC passes, diagnostics.
# Everything goes to gpurun_out/final/; copy what is to be judged into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
step() { echo "== $*" >&2; }
step pmc traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/gemm_traffic.json > /dev/null || exit 1
cp $O/gemm_traffic.json $R/profiles/r03_gemm_traffic.json      # bench.py reports roofline.traffic from the profile of the same kernel sources
step bench
python3 $R/bench.py --steps 20 --warmup 5 --gemm-shapes $O/nt_shapes.txt > $O/bench_cfg2.json 2>$O/bench_cfg2.err || exit 1
python3 $R/bench.py --aux --no-cpu-baseline > $O/bench_cfg3_aux.json 2>/dev/null || exit 1
python3 $R/bench.py --config 4 --no-cpu-baseline > $O/bench_cfg4_224.json 2>/dev/null || exit 1
python3 $R/bench.py --config 4 --image 336 --no-cpu-baseline > $O/bench_cfg4_336.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype f16 --no-cpu-baseline > $O/bench_cfg2_f16.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype bf16x3 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_cfg2_bf16x3.json 2>/dev/null || exit 1
python3 $R/bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null || exit 1
python3 $R/bench.py --config 5 --aux --no-cpu-baseline > $O/bench_cfg5_aux.json 2>/dev/null || exit 1
step kernel stats
rocprofv3 --kernel-trace --stats -d /tmp/ks_on -o ks --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
cp $(find /tmp/ks_on -name "*kernel_stats.csv" | head -1) $O/kernel_stats_on.csv
python3 $R/tools/timeline.py /tmp/ks_on 6 > $O/timeline.txt
export MMHIP_OVERLAP=0
rocprofv3 --kernel-trace --stats -d /tmp/ks_off -o ks --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
unset MMHIP_OVERLAP
cp $(find /tmp/ks_off -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial.csv
python3 $R/tools/hbm_table.py $O/kernel_stats_serial.csv > $O/hbm_kernels.md
step pmc mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_m -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_mfma.py /tmp/pmc_m > $O/mfma_busy.txt
step diagnostics
cd $R
python3 tools/vendor_gemm_bench.py > $O/vendor_gemm.txt 2>/dev/null
VARIANTS=1,9,13,15,16,18 python3 tools/gemm8_bench.py > $O/gemm8_microbench.txt 2>/dev/null
{ tools/pmc_clock.sh 8192 8192 8192 15 8; tools/pmc_clock.sh 16384 3072 3072 15 12; tools/pmc_clock.sh 12608 2304 768 15 30; } > $O/gemm_clock.txt 2>/dev/null
{ tools/pmc_l2.sh 12608 2304 768 15 5 cold; tools/pmc_l2.sh 12608 2304 768 1 5 cold; tools/pmc_l2.sh 8192 8192 8192 15 3; } > $O/l2_hit.txt 2>/dev/null
{ tools/ab_env.sh "BASE=1" "MMHIP_PART=0" "MMHIP_PART=128,128" "MMHIP_EARLY_ADAMW=0" "MMHIP_VIT_PRIO=0" "MMHIP_OVERLAP=0" "MMHIP_DETERMINISTIC=1"; BENCH_ARGS="--config 3" tools/ab_env.sh "BASE=1" "MMHIP_PART=0" "MMHIP_EARLY_ADAMW=0"; BENCH_ARGS="--config 4" tools/ab_env.sh "BASE=1" "MMHIP_PART=0"; } > $O/step_ab.txt 2>/dev/null
python3 -m pytest tests/test_gpu_model.py -q -s -k "train_losses_and_grads or dropout_train_step or forward_matches or config4 or eval_loop" 2>&1 | grep -v "^$" > $O/parity.txt
echo done >&2
