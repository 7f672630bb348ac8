set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_mshds_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
timeout -k 10 900 python bench.py --config C2 --no-cpu-baseline --no-inclusive > gpurun_out/r02/bench_C2_x.json 2> gpurun_out/r02/bench_C2_x.err || { tail -5 gpurun_out/r02/bench_C2_x.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_C2_x.json').read().strip().splitlines()[-1])
k=d['kernels']
print(d['value'], d['ms_per_step'], 'cc', k['mshds_pitch_cc_frames']['ms'], 'ac', k['mshds_pitch_ac_frames']['ms'])
PY
done
