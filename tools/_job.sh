set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for v in 1 0 1 0; do
RSAF_EXP_UNBALANCED=$v timeout -k 10 600 python bench.py --config C3 --no-cpu-baseline --no-inclusive > gpurun_out/r02/bench_C3_x.json 2> gpurun_out/r02/bench_C3_x.err || { tail -5 gpurun_out/r02/bench_C3_x.err; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/r02/bench_C3_x.json').read().strip().splitlines()[-1])
k=d['kernels']
print("unbalanced=$v", d['value'], d['ms_per_step'], 'gemm', k['w2v2_gemm']['ms'], k['w2v2_gemm']['launches'], 'ln', k['w2v2_layernorm']['ms'], 'attn', k['w2v2_attn_fused']['ms'], 'pos', k['w2v2_posconv_gemm']['ms'])
PY
done
