"""Known answers for the CPU restatements of the two resamplers (no GPU)."""
import numpy as np

from oracle import resample_oracle as ro


def test_sinc_hann_lengths_and_passband():
    for fs, n in ((44100, 4410), (48000, 4801), (8000, 801), (22050, 2205)):
        x = np.zeros(n, np.float32)
        assert len(ro.resample_sinc_hann(x, fs, 16000)) == -(-16000 * n // fs)      # ceil, like torchaudio
    fs = 44100
    t = np.arange(fs // 2) / fs
    x = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    y = ro.resample_sinc_hann(x, fs, 16000)
    ref = np.sin(2 * np.pi * 1000.0 * np.arange(len(y)) / 16000.0)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 2e-3                        # sample i*orig/new of the input grid
    z = ro.resample_sinc_hann(np.sin(2 * np.pi * 12000.0 * t).astype(np.float32), fs, 16000)
    assert np.abs(z[200:-200]).max() < 2e-3                                         # above the new Nyquist: removed


def test_sinc_hann_kernel_is_a_partition_of_unity_at_dc():
    kern, width, o, n = ro.sinc_hann_kernel(44100, 16000)
    assert kern.shape == (160, 2 * width + 441)
    assert np.abs(kern.sum(axis=1) - 1.0).max() < 2e-3                              # DC gain of every phase


def test_praat_resample_grid_and_passband():
    fs = 44100.0
    n = 22050
    t = (np.arange(n) + 0.5) / fs                                                  # Praat: first sample at half a period
    x = np.sin(2 * np.pi * 700.0 * t)
    y = ro.resample_praat(x, fs, 16000.0, 50)
    m = int(np.floor(n / fs * 16000.0 + 0.5))
    assert len(y) == m
    to = 0.5 * (n / fs - (m - 1) / 16000.0) + np.arange(m) / 16000.0
    assert np.abs(y[100:-100] - np.sin(2 * np.pi * 700.0 * to)[100:-100]).max() < 2e-3
    up = ro.resample_praat(x[:2000], 8000.0, 16000.0, 50)                           # upsampling keeps the full band
    assert len(up) == 4000
