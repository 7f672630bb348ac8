set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02/final
timeout -k 10 900 python bench.py --config e2e > gpurun_out/r02/final/bench_e2e.json 2> gpurun_out/r02/final/bench_e2e.err || { tail -5 gpurun_out/r02/final/bench_e2e.err; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/r02/final/bench_e2e.json').read().strip().splitlines()[-1])
print("e2e", d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'), d['roofline'].get('fp32_equivalent_tflops'), (d.get('inclusive_of_pcie_and_decode') or {}).get('value'), d['cpu_baseline'].get('value'), d['roofline'].get('traffic'))
PY
