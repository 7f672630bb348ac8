set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
P="python3 bench.py --no-cpu-baseline --no-inclusive"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/stats_e2e --output-format csv -- $P > gpurun_out/r02/bench_e2e_under_rocprof.json 2> gpurun_out/r02/stats.err
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02/pmc_fetch --output-format csv -- $P > gpurun_out/r02/pmc_fetch.json 2> gpurun_out/r02/pmc_fetch.err
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02/pmc_write --output-format csv -- $P > gpurun_out/r02/pmc_write.json 2> gpurun_out/r02/pmc_write.err
echo write done
python tools/pmc_traffic.py gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write gpurun_out/r02/pmc_bench_traffic.json --config e2e --clips 1000 --windows 2048 --command "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- $P" | head -20
ls gpurun_out/r02/stats_e2e/*/ | head
# keep only the summaries (the traces are large)
find gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write -name "*counter_collection.csv" -size +20M -delete || true
find gpurun_out/r02/stats_e2e -name "*kernel_trace.csv" -size +20M -delete || true
