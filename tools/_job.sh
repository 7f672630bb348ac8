set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/final3
export TMPDIR=/tmp
O=gpurun_out/r02/final3
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1 || (tail -30 $O/gputests.log; exit 1)
tail -2 $O/gputests.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-inclusive > $O/bench_default_under_rocprof.json 2> $O/rocprof.err
cp "$(find $O/prof -name '*kernel_stats.csv' -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2)" $O/bench_default_kernel_stats.csv
rm -f $O/prof/*kernel_trace.csv
echo rocprof done
timeout -k 10 500 python bench.py > $O/bench_e2e.json 2> $O/bench_e2e.err
echo e2e done
timeout -k 10 400 python bench.py --config C2 > $O/bench_C2.json 2> $O/bench_C2.err
echo C2 done
timeout -k 10 400 python bench.py --config C3 > $O/bench_C3.json 2> $O/bench_C3.err
echo C3 done
timeout -k 10 300 python bench.py --config C4 > $O/bench_C4.json 2> $O/bench_C4.err
echo C4 done
