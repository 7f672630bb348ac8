set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/rs
timeout -k 10 600 python -m pytest tests/test_mshds_gpu.py tests/test_resample_gpu.py -q -x > gpurun_out/r02/rs/test5.log 2>&1 || (tail -30 gpurun_out/r02/rs/test5.log; exit 1)
tail -2 gpurun_out/r02/rs/test5.log
timeout -k 10 300 python bench.py --config C2 --no-cpu-baseline --no-inclusive > gpurun_out/r02/rs/bench_C2d.json 2> gpurun_out/r02/rs/bench_C2d.err || (tail -5 gpurun_out/r02/rs/bench_C2d.err; exit 1)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/rs/bench_C2d.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in d['kernels'].items(): print(k, v['launches'], v['ms'])
PY
