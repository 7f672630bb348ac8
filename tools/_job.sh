set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/rs
timeout -k 10 900 python -m pytest tests/test_resample_gpu.py tests/test_mshds_gpu.py -q -x > gpurun_out/r02/rs/test2.log 2>&1
