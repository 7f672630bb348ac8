set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_mshds_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 600 python tools/pitch_phase.py 64 > gpurun_out/r02/pitch_phase_fft.txt 2>&1 || { tail -20 gpurun_out/r02/pitch_phase_fft.txt; exit 1; }
grep "^cc" gpurun_out/r02/pitch_phase_fft.txt | grep "stop 0\|stop 1 \|stop 2"
