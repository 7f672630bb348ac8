set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t7.log 2>&1 || { tail -60 gpurun_out/r02/t7.log; exit 1; }
tail -3 gpurun_out/r02/t7.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_e2e_split6.json 2> gpurun_out/r02/bench_e2e_split6.err || { tail -30 gpurun_out/r02/bench_e2e_split6.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_e2e_split6.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['fp32_equivalent_tflops'], d['roofline']['frac'], d['inclusive_of_pcie_and_decode'])
PY
