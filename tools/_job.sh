set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 600 python3 tools/gemm6_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-215
G6_ALL_A_PANELS=1 timeout -k 10 600 python3 tools/gemm6_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-215
