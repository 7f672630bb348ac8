set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02/final
for cfg in e2e C2 C3 C4; do
  timeout -k 10 900 python bench.py --config $cfg > gpurun_out/r02/final/bench_$cfg.json 2> gpurun_out/r02/final/bench_$cfg.err || { tail -5 gpurun_out/r02/final/bench_$cfg.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r02/final/bench_$cfg.json').read().strip().splitlines()[-1])
print("$cfg", d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'), d['roofline'].get('fp32_equivalent_tflops'), (d.get('inclusive_of_pcie_and_decode') or {}).get('value'), d['cpu_baseline'].get('value'), d['roofline'].get('traffic'))
PY
done
