set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k bf16x6 > gpurun_out/r02/t5.log 2>&1 || { tail -40 gpurun_out/r02/t5.log; exit 1; }
tail -2 gpurun_out/r02/t5.log
timeout -k 10 600 python tools/gemm6_bench.py > gpurun_out/r02/gemm6_bench_pf.txt 2>&1 || { tail -20 gpurun_out/r02/gemm6_bench_pf.txt; exit 1; }
grep -v "amdgpu.ids\|conv1" gpurun_out/r02/gemm6_bench_pf.txt | cut -c1-125
