set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_cnnlstm_gpu.py tests/test_mshds_gpu.py -x -q -m gpu -k "stage or standalone or c4 or c1 or c2 or logits_match" > gpurun_out/r02/t1.log 2>&1 || { tail -30 gpurun_out/r02/t1.log; exit 1; }
tail -3 gpurun_out/r02/t1.log
rocprofv3 -L > gpurun_out/r02/counters.txt 2>&1 || true
P="python3 bench.py --stages smile --no-cpu-baseline --steps 3 --warmup 1 --clips 512"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/r02/pmc_smile_a --output-format csv -- $P > gpurun_out/r02/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d gpurun_out/r02/pmc_smile_b --output-format csv -- $P > gpurun_out/r02/pmc_b.log 2>&1
python tools/pmc_summary.py gpurun_out/r02/pmc_smile_before.json smile gpurun_out/r02/pmc_smile_a gpurun_out/r02/pmc_smile_b > /dev/null
tail -2 gpurun_out/r02/pmc_a.log
