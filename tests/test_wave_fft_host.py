"""csrc/wave_fft.h replayed on the host (tests/host/wave_fft_replay.cpp): the lane functions of the one-wave-per-frame
transform of the MSHDS pitch kernels run for an emulated wavefront; the replay checks the arithmetic of the transform and of
the autocorrelation / cross-correlation chains against direct sums, and that no LDS access has a bank conflict."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_wave_fft_replay(tmp_path):
    exe = str(tmp_path / "wave_fft_replay")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "robust_speech_analysis_framework_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "wave_fft_replay.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(lines) == 8 and all(l.startswith("ok ") for l in lines), r.stdout
