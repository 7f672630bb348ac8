"""Device resamplers in front of the extractors vs their CPU restatements (parity unpinned: torchaudio/Praat absent)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import resample_oracle as ro
from robust_speech_analysis_framework_amd import synth


def _clip(fs, seconds, k=300):
    return synth.synth_clip(k, seconds, fs=fs)


@pytest.mark.parametrize("fs", [44100, 48000, 22050, 8000, 11025])
def test_sinc_hann_matches_torchaudio_restatement(rsaf_lib, fs):
    from robust_speech_analysis_framework_amd.resample import resample_sinc_hann
    x = _clip(fs, 0.7)
    got = resample_sinc_hann(x, fs, 16000).cpu().numpy()
    ref = ro.resample_sinc_hann(x, fs, 16000)
    assert got.shape == ref.shape                                     # ceil(new * n / orig), exact
    assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_sinc_hann_edges_and_identity(rsaf_lib):
    from robust_speech_analysis_framework_amd.resample import resample_sinc_hann
    x = _clip(16000, 0.1)
    assert np.array_equal(resample_sinc_hann(x, 16000, 16000).cpu().numpy(), x)
    for n in (1, 2, 441, 442, 1000):                                  # shorter than one polyphase frame and ragged tails
        xs = _clip(44100, 0.05)[:n]
        got = resample_sinc_hann(xs, 44100, 16000).cpu().numpy()
        ref = ro.resample_sinc_hann(xs, 44100, 16000)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-6


@pytest.mark.parametrize("fs", [44100, 48000, 32000, 22050, 8000, 11025])
def test_praat_resample_matches_restatement(rsaf_lib, fs):
    from robust_speech_analysis_framework_amd.resample import resample_praat
    x = _clip(fs, 0.6, k=301)
    got = resample_praat(x, fs, 16000, 50).cpu().numpy()
    ref = ro.resample_praat(x, float(fs), 16000.0, 50)
    assert got.shape == ref.shape                                     # round(n / fs * 16000), exact
    assert np.abs(got - ref).max() <= 2e-7 * max(1.0, np.abs(ref).max())   # float32 output of float64 sums


@pytest.mark.parametrize("fs,n,depth", [(44100, 1, 50), (44100, 2, 50), (44100, 7, 50), (48000, 47, 50), (48000, 49, 3), (44100, 2096, 50),
                                        (44100, 2097, 50), (22050, 30721, 50), (48000, 260145, 50), (44100, 1046577, 500),
                                        (24000, 5000, 1), (20000, 5000, 2), (32000, (1 << 24) - 1000, 2), (22050, (1 << 25) + 5, 1)])
def test_praat_resample_sizes_and_depths(rsaf_lib, fs, n, depth):
    """Every shape of the four-step transform (nfft 2^11 ... 2^21: 2 ... 512 rows), the clipped interpolation depths at the
    edges (nearest / linear / cubic), odd and even first-cleared positions; the two longest transforms (2^25 and 2^26
    samples: rows of 4 096, columns of 4 096 / 8 192 points in 128 KB of LDS) with 8.7 and 12.7 minutes of sound."""
    from robust_speech_analysis_framework_amd.resample import resample_praat
    rng = np.random.Generator(np.random.PCG64(n))
    x = (0.3 * rng.standard_normal(n)).astype(np.float32)             # white: energy at every cleared and kept bin
    got = resample_praat(x, fs, 16000, depth).cpu().numpy()
    ref = ro.resample_praat(x, float(fs), 16000.0, depth)
    assert got.shape == ref.shape                                     # (1 sample at 44.1 kHz -> 0 samples: Praat refuses, empty here)
    assert len(ref) == 0 or np.abs(got - ref).max() <= 3e-7, np.abs(got - ref).max()


def test_praat_lowpass_batch_of_ragged_sounds_matches_restatement(rsaf_lib):
    """The batched low-pass behind To Formant (burg): sounds of different transform lengths in one launch (the shorter ones
    stride through the twiddle tables of the longest), packed back to back like the clips of the MSHDS engine."""
    import torch
    from robust_speech_analysis_framework_amd import _lib
    from robust_speech_analysis_framework_amd.mshds import LP_SIG, _dev
    lib = _lib.load()
    lengths = [49, 2049, 6193, 30000, 14385, 140001, 1]
    rng = np.random.Generator(np.random.PCG64(17))
    x = (0.4 * rng.standard_normal(sum(lengths))).astype(np.float32)
    sigs = np.zeros(len(lengths), dtype=LP_SIG)
    off = work = 0
    for i, n in enumerate(lengths):
        lg = max(11, int(n + 2000 - 1).bit_length())
        assert (1 << lg) >= n + 2000 and (lg == 11 or (1 << (lg - 1)) < n + 2000)
        sigs[i] = (off, off, work, n, lg)
        off += n
        work += 1 << (lg - 1)
    xd = torch.from_numpy(x).cuda()
    out = torch.full((len(x),), float("nan"), dtype=torch.float64, device="cuda")
    wk = torch.empty(2 * work, dtype=torch.float64, device="cuda")
    _lib.check(lib.rsaf_praat_lowpass_batch(_lib.ptr(xd), _lib.ptr(_dev(sigs, "cuda")), len(lengths), int(sigs["lg"].max()), 0.625,
                                            _lib.ptr(wk), work, _lib.ptr(out), _lib.stream_ptr(None)), "rsaf_praat_lowpass_batch")
    got = out.cpu().numpy()
    off = 0
    for n in lengths:
        ref = ro.praat_fft_lowpass(x[off:off + n].astype(np.float64), 0.625)
        assert np.abs(got[off:off + n] - ref).max() <= 1e-13, n
        off += n


@pytest.mark.parametrize("n", [4_800_001, 9_000_000])
def test_praat_lowpass_of_a_five_and_a_nine_minute_sound(rsaf_lib, n):
    """The two longest transform shapes: 2^23 samples (2 048 x 2 048 complex points, two columns per workgroup) and 2^24
    (4 096-point column transforms, one column per workgroup)."""
    import torch
    from robust_speech_analysis_framework_amd import _lib
    from robust_speech_analysis_framework_amd.mshds import LP_SIG, _dev
    lib = _lib.load()
    rng = np.random.Generator(np.random.PCG64(n))
    x = (0.4 * rng.standard_normal(n)).astype(np.float32)
    lg = int(n + 2000 - 1).bit_length()
    sigs = np.zeros(1, dtype=LP_SIG)
    sigs[0] = (0, 0, 0, n, lg)
    xd = torch.from_numpy(x).cuda()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    wk = torch.empty(2 << (lg - 1), dtype=torch.float64, device="cuda")
    _lib.check(lib.rsaf_praat_lowpass_batch(_lib.ptr(xd), _lib.ptr(_dev(sigs, "cuda")), 1, lg, 0.625, _lib.ptr(wk), 1 << (lg - 1),
                                            _lib.ptr(out), _lib.stream_ptr(None)), "rsaf_praat_lowpass_batch")
    ref = ro.praat_fft_lowpass(x.astype(np.float64), 0.625)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-12


def test_praat_resample_rejects_missing_workspace(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd import _lib
    lib = _lib.load()
    x = torch.zeros(4000, device="cuda")
    out = torch.zeros(2000, device="cuda")
    assert lib.rsaf_resample_praat_work_bytes(4000, 32000.0, 16000.0) == 8192 * 8 + 4000 * 8
    assert lib.rsaf_resample_praat_work_bytes(4000, 11025.0, 16000.0) == 0
    assert lib.rsaf_resample_praat_work_bytes(4000, 8000.0, 16000.0) == 8192 * 8 + 2 * 4000 * 8      # Sound_upsample
    rc = lib.rsaf_resample_praat(_lib.ptr(x), 4000, 32000.0, 16000.0, 50, _lib.ptr(out), 2000, None, 0, _lib.stream_ptr(None))
    assert rc != 0 and b"workspace" in lib.rsaf_last_error()


def test_mshds_dropin_accepts_44k1_files(rsaf_lib, tmp_path):
    import pandas as pd
    from robust_speech_analysis_framework_amd.mshds import FEATURE_NAMES, extract_mshds_features
    from oracle import mshds_oracle as mo
    pcm = synth.synth_clip_int16(302, 1.5, fs=44100)
    path = str(tmp_path / "clip44k.wav")
    synth.write_wav(path, pcm, fs=44100)
    df = extract_mshds_features(pd.DataFrame({"filepath": [path]}), verbose=False)
    assert list(df.columns) == ["filename"] + FEATURE_NAMES and len(df) == 1
    x16, x1, xmax = ro.resample_praat_sound(pcm.astype(np.float32) / np.float32(32768.0), 44100.0, 16000.0, 50)
    ref, _ = mo.extract(x16, x1, xmax)
    got = df[FEATURE_NAMES].to_numpy(dtype=np.float64)[0]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert (np.abs(got[ok] - ref[ok]) <= 1e-4 * np.maximum(np.abs(ref[ok]), 1e-3)).all(), (got, ref)


@pytest.mark.parametrize("width,nch", [(2, 1), (2, 2), (1, 2), (3, 1), (4, 3)])
def test_pcm_decode_and_mixdown_on_device_is_bit_identical_to_the_host_reader(rsaf_lib, tmp_path, width, nch):
    import wave
    from robust_speech_analysis_framework_amd.wavio import read_wav_mono, read_wav_mono_device
    rng = np.random.Generator(np.random.PCG64(7 + width + nch))
    n = 4097
    if width == 1:
        raw = rng.integers(0, 256, size=(n, nch), dtype=np.uint8).tobytes()
    elif width == 2:
        raw = rng.integers(-32768, 32768, size=(n, nch), dtype=np.int16).astype("<i2").tobytes()
    elif width == 3:
        v = rng.integers(-(1 << 23), 1 << 23, size=(n, nch), dtype=np.int32)
        b = np.stack([(v & 0xFF), ((v >> 8) & 0xFF), ((v >> 16) & 0xFF)], axis=-1).astype(np.uint8)
        raw = b.tobytes()
    else:
        raw = rng.integers(-(1 << 31), 1 << 31, size=(n, nch), dtype=np.int64).astype("<i4").tobytes()
    path = str(tmp_path / f"pcm{width}_{nch}.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(nch); w.setsampwidth(width); w.setframerate(22050); w.writeframes(raw)
    host, fs = read_wav_mono(path)
    dev, fs2, nfr = read_wav_mono_device(path)
    assert fs == fs2 == 22050 and nfr == n == len(host)
    assert np.array_equal(dev.cpu().numpy(), host)                      # same float32 operations in the same order


def test_praat_resampled_sound_keeps_its_time_axis(rsaf_lib):
    """Known answer: a windowed tone resampled from 44.1 kHz keeps its phase at t = 0 when the samples are read on the
    axis (x1, dx) that comes back with them; read as if x1 were dx / 2 the phase is off by x1 - dx / 2.  A doubling of the
    rate is Praat's Sound_upsample: sample 2p IS input sample p (filtered), declared a quarter input period earlier."""
    from robust_speech_analysis_framework_amd.resample import resample_praat_sound
    f0, ph = 440.0, 0.7

    def phase_at_zero(y, x1):
        m = len(y)
        ty = x1 + np.arange(m) / 16000.0
        sl = slice(m // 4, 3 * m // 4)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * ty[sl] / 0.5)
        A = np.stack([np.sin(2 * np.pi * f0 * ty[sl]), np.cos(2 * np.pi * f0 * ty[sl])], 1) * w[:, None]
        c = np.linalg.lstsq(A, y[sl].astype(np.float64), rcond=None)[0]
        return np.arctan2(c[1], c[0])
    for fs in (44100.0, 48000.0, 22050.0):
        n = int(fs * 0.5) + 7                                          # 0.5 s and a bit: the new grid is off the file grid
        t = (np.arange(n) + 0.5) / fs
        x = (np.sin(2 * np.pi * f0 * t + ph) * (0.5 - 0.5 * np.cos(2 * np.pi * t / 0.5))).astype(np.float32)
        y, x1, xmax = resample_praat_sound(x, fs, 16000.0, 50)
        yo, x1o, xmaxo = ro.resample_praat_sound(x, fs, 16000.0, 50)
        assert x1 == x1o and xmax == xmaxo == n / fs and len(y) == len(yo)
        assert abs(x1 - 0.5 / 16000.0) > 1e-7
        y = y.cpu().numpy()
        assert abs(phase_at_zero(y, x1) - ph) / (2 * np.pi * f0) <= 1e-9                  # seconds
        assert abs(abs(phase_at_zero(y, 0.5 / 16000.0) - ph) / (2 * np.pi * f0) - abs(x1 - 0.5 / 16000.0)) <= 1e-9
    n = 4000
    t = (np.arange(n) + 0.5) / 8000.0
    x = (np.sin(2 * np.pi * f0 * t + ph) * (0.5 - 0.5 * np.cos(2 * np.pi * t / 0.5))).astype(np.float32)
    y, x1, xmax = resample_praat_sound(x, 8000.0, 16000.0, 50)
    assert len(y) == 2 * n and x1 == 0.5 / 8000.0 - 0.25 / 8000.0 and xmax == 0.5
    y = y.cpu().numpy()
    assert np.abs(y[0::2][100:-100] - x[100:-100]).max() <= 1e-5                           # 440 Hz lies far below the ramp
    assert abs((phase_at_zero(y, x1) - ph) / (2 * np.pi * f0) - 0.25 / 8000.0) <= 1e-8       # Praat's labelling, kept


def test_praat_resample_refuses_sounds_beyond_the_transform(rsaf_lib):
    """2^26 - 2000 samples is the longest sound the low-pass takes; one more sample is an argument error (no launch)."""
    import torch
    from robust_speech_analysis_framework_amd import _lib
    lib = _lib.load()
    n = int(lib.rsaf_praat_lowpass_max_samples()) + 1
    assert lib.rsaf_resample_praat_work_bytes(n, 44100.0, 16000.0) > 0
    x = torch.zeros(16, device="cuda")                                  # never touched: the size check comes first
    rc = lib.rsaf_resample_praat(_lib.ptr(x), n, 44100.0, 16000.0, 50, _lib.ptr(x), 1, _lib.ptr(x), 1 << 40, _lib.stream_ptr(None))
    assert rc != 0 and b"2^26" in lib.rsaf_last_error()
