set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_smile_gpu.py -x -q -m gpu > gpurun_out/r02/t2.log 2>&1 || { tail -60 gpurun_out/r02/t2.log; exit 1; }
tail -5 gpurun_out/r02/t2.log
