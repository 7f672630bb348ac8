set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_smile_gpu.py -x -q -m gpu > gpurun_out/r02/t4.log 2>&1 || { tail -60 gpurun_out/r02/t4.log; exit 1; }
tail -3 gpurun_out/r02/t4.log
python bench.py --stages smile --no-cpu-baseline --steps 3 --warmup 1 --clips 1000 > gpurun_out/r02/bench_smile_v5.json 2> gpurun_out/r02/bench_smile_v5.err || { tail -20 gpurun_out/r02/bench_smile_v5.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_smile_v5.json').read())
print(d['ms_per_step'], d['kernels'])
PY
