set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/r02/clk --output-format csv -- python3 tools/gemm6_bench.py > gpurun_out/r02/clk.log 2>&1 || { tail -5 gpurun_out/r02/clk.log; exit 1; }
python tools/micro/gemm_clock.py gpurun_out/r02/clk | tee gpurun_out/r02/gemm_clock.txt
head -2 $(find gpurun_out/r02/clk -name "*kernel_trace.csv" | head -1) | cut -c1-300
find gpurun_out/r02/clk -name "*.csv" -size +1M -delete || true
