set -e
python -m pytest tests/test_gpu_ops.py -x -q -k gemm_nt > gpurun_out/t13.log 2>&1
for r in 2 7 8; do
  echo "rule $r" >> gpurun_out/ab13.log
  MMHIP_NT_RULE=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['fwd_bwd_ms'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])" >> gpurun_out/ab13.log
done
for r in 2 7 8; do
  echo "rule $r again" >> gpurun_out/ab13.log
  MMHIP_NT_RULE=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['fwd_bwd_ms'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])" >> gpurun_out/ab13.log
done
