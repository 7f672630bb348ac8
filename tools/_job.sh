set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_mshds_gpu.py -m gpu -q 2>&1 | tail -5
timeout -k 10 600 python tests/sweeps/mshds_fuzz.py 7000 24 > gpurun_out/r02/mshds_fuzz_lazy.log 2>&1 || true
grep -c "MISMATCH" gpurun_out/r02/mshds_fuzz_lazy.log || true
tail -2 gpurun_out/r02/mshds_fuzz_lazy.log
timeout -k 10 600 python tools/pitch_phase.py 64 > gpurun_out/r02/pitch_phase_fft.txt 2>&1 || { tail -20 gpurun_out/r02/pitch_phase_fft.txt; exit 1; }
grep "^cc_hnr" gpurun_out/r02/pitch_phase_fft.txt | grep "stop 0\|stop 5"
