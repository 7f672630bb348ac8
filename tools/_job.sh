set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 500 python tests/sweeps/stage_fuzz.py 8100 16 > gpurun_out/r02/stage_fuzz_r02.log 2>&1 || { tail -30 gpurun_out/r02/stage_fuzz_r02.log; exit 1; }
tail -4 gpurun_out/r02/stage_fuzz_r02.log
FUZZ_MIN_S=3 FUZZ_MAX_S=9 timeout -k 10 700 python tests/sweeps/mshds_fuzz.py 7300 24 > gpurun_out/r02/mshds_fuzz_long_r02.log 2>&1 || { tail -30 gpurun_out/r02/mshds_fuzz_long_r02.log; exit 1; }
tail -3 gpurun_out/r02/mshds_fuzz_long_r02.log
