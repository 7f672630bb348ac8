set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_mshds_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 500 python tests/sweeps/mshds_edge.py > gpurun_out/r02/mshds_edge_r02.log 2>&1 || { tail -30 gpurun_out/r02/mshds_edge_r02.log; exit 1; }
tail -2 gpurun_out/r02/mshds_edge_r02.log
timeout -k 10 600 python tests/sweeps/mshds_fuzz.py 7100 24 > gpurun_out/r02/mshds_fuzz_r02.log 2>&1 || { tail -30 gpurun_out/r02/mshds_fuzz_r02.log; exit 1; }
tail -2 gpurun_out/r02/mshds_fuzz_r02.log
timeout -k 10 900 python bench.py --config C2 --no-cpu-baseline --no-inclusive > gpurun_out/r02/bench_C2_x.json 2> gpurun_out/r02/bench_C2_x.err || { tail -5 gpurun_out/r02/bench_C2_x.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_C2_x.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:6]: print(k, v)
PY
