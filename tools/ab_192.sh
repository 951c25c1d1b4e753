set -e
for v in 1 0 1 0; do
  rm -f gpurun_out/shapes_$v.txt
  MMHIP_NT8_192=$v python bench.py --no-cpu-baseline --gemm-shapes gpurun_out/shapes_$v.txt > gpurun_out/bench_192_$v.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_192_$v.json")); print("NT8_192=$v", d["ms_per_step"], d["fwd_bwd_ms"], d["roofline"]["frac"], d["roofline"]["frac_serial"])
PY
done
