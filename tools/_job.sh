set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/final3
timeout -k 10 100 python -m pytest tests/test_resample_gpu.py tests/test_mshds_gpu.py -q -x -k "resample or formants or cpp or lowpass" > gpurun_out/r02/final3/last_check.log 2>&1 || (tail -20 gpurun_out/r02/final3/last_check.log; exit 1)
tail -2 gpurun_out/r02/final3/last_check.log
