set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_w2v2_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout -k 10 600 python bench.py --config C3 --no-cpu-baseline --no-inclusive > gpurun_out/r02/bench_C3_x.json 2> gpurun_out/r02/bench_C3_x.err || { tail -5 gpurun_out/r02/bench_C3_x.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_C3_x.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['fp32_equivalent_tflops'], d['roofline']['frac'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:8]: print(k, v)
PY
