set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 500 python tests/sweeps/mshds_edge.py > gpurun_out/r02/mshds_edge_r02.log 2>&1 || { tail -30 gpurun_out/r02/mshds_edge_r02.log; exit 1; }
tail -8 gpurun_out/r02/mshds_edge_r02.log
timeout -k 10 600 python tests/sweeps/mshds_fuzz.py 7000 24 > gpurun_out/r02/mshds_fuzz_r02.log 2>&1 || { tail -30 gpurun_out/r02/mshds_fuzz_r02.log; exit 1; }
tail -4 gpurun_out/r02/mshds_fuzz_r02.log
