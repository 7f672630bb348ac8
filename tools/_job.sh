set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/rs
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/r02/rs/kt -o kt --output-format csv -- python3 bench.py --config C2 --no-cpu-baseline --no-inclusive --steps 1 --warmup 1 > gpurun_out/r02/rs/kt_bench.json 2> gpurun_out/r02/rs/kt_bench.err
ls -la gpurun_out/r02/rs/kt | head
