set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/rs
timeout -k 10 500 python -m pytest tests/test_resample_gpu.py -q > gpurun_out/r02/rs/test.log 2>&1
