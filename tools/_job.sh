set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_cache_formats_gpu.py -x -q -m gpu > gpurun_out/r02/t8.log 2>&1 || { tail -60 gpurun_out/r02/t8.log; exit 1; }
tail -3 gpurun_out/r02/t8.log
