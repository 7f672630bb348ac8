set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r02/final; mkdir -p gpurun_out/r02/final
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
P="python3 bench.py --no-cpu-baseline --no-inclusive"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02/final/pmc_fetch --output-format csv -- $P > gpurun_out/r02/final/pmc_fetch.json 2> gpurun_out/r02/final/pmc_fetch.err || { tail -5 gpurun_out/r02/final/pmc_fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02/final/pmc_write --output-format csv -- $P > gpurun_out/r02/final/pmc_write.json 2> gpurun_out/r02/final/pmc_write.err || { tail -5 gpurun_out/r02/final/pmc_write.err; exit 1; }
python tools/pmc_traffic.py gpurun_out/r02/final/pmc_fetch gpurun_out/r02/final/pmc_write profiles/r02/pmc_bench_traffic.json --config e2e --clips 1000 --windows 2048 --shape-windows 1752 --command "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- $P" | head -12
cp profiles/r02/pmc_bench_traffic.json gpurun_out/r02/final/pmc_bench_traffic.json
find gpurun_out/r02/final -name "*counter_collection.csv" -delete || true
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/final/stats_e2e --output-format csv -- $P > gpurun_out/r02/final/bench_default_under_rocprof.json 2> gpurun_out/r02/final/stats.err || { tail -5 gpurun_out/r02/final/stats.err; exit 1; }
cp $(find gpurun_out/r02/final/stats_e2e -name "*kernel_stats.csv" | head -1) gpurun_out/r02/final/bench_default_kernel_stats.csv
find gpurun_out/r02/final -name "*kernel_trace.csv" -delete || true
for cfg in e2e C2 C3 C4; do
  timeout -k 10 900 python bench.py --config $cfg > gpurun_out/r02/final/bench_$cfg.json 2> gpurun_out/r02/final/bench_$cfg.err || { tail -5 gpurun_out/r02/final/bench_$cfg.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r02/final/bench_$cfg.json').read().strip().splitlines()[-1])
print("$cfg", d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'), d['roofline'].get('fp32_equivalent_tflops'), (d.get('inclusive_of_pcie_and_decode') or {}).get('value'), d['cpu_baseline'].get('value'), d['cpu_baseline']['one_thread']['value'], d['roofline'].get('traffic'), d['roofline'].get('avg_launch_ms'), d['roofline'].get('launches'))
PY
done
