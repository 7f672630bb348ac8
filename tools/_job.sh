set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/sweeps2
timeout -k 10 300 python -m pytest tests/test_resample_gpu.py -q -x -k "five_and_a_nine or ragged" > gpurun_out/r02/sweeps2/long_lowpass.log 2>&1 || (tail -30 gpurun_out/r02/sweeps2/long_lowpass.log; exit 1)
tail -2 gpurun_out/r02/sweeps2/long_lowpass.log
timeout -k 10 400 python tests/sweeps/long_clip.py > gpurun_out/r02/sweeps2/long_clip.log 2>&1 || (tail -30 gpurun_out/r02/sweeps2/long_clip.log; exit 1)
tail -4 gpurun_out/r02/sweeps2/long_clip.log
