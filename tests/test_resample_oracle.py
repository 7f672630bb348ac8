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


def _interpolate_sinc_scalar(y, x, depth):
    """NUM_interpolate_sinc written out sample by sample on Praat's 1-based index (the loop form of the published source)."""
    n = len(y)
    midleft = int(np.floor(x)); midright = midleft + 1
    if x > n: return y[n - 1]
    if x < 1: return y[0]
    if x == midleft: return y[midleft - 1]
    md = min(depth, midright - 1, n - midleft)
    if md <= 0: return y[int(np.floor(x + 0.5)) - 1]
    if md == 1: return y[midleft - 1] + (x - midleft) * (y[midright - 1] - y[midleft - 1])
    if md == 2:
        yl, yr = y[midleft - 1], y[midright - 1]
        dyl, dyr = 0.5 * (yr - y[midleft - 2]), 0.5 * (y[midright] - yl)
        fil, fir = x - midleft, midright - x
        return yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2 * (yr - yl)))
    left, right = midright - md, midleft + md
    res = 0.0
    a = np.pi * (x - midleft); halfsina = 0.5 * np.sin(a); aa = a / (x - left + 1.0); daa = np.pi / (x - left + 1.0)
    for ix in range(midleft, left - 1, -1):
        res += y[ix - 1] * (halfsina / a * (1.0 + np.cos(aa)))
        a += np.pi; aa += daa; halfsina = -halfsina
    a = np.pi * (midright - x); halfsina = 0.5 * np.sin(a); aa = a / (right - x + 1.0); daa = np.pi / (right - x + 1.0)
    for ix in range(midright, right + 1):
        res += y[ix - 1] * (halfsina / a * (1.0 + np.cos(aa)))
        a += np.pi; aa += daa; halfsina = -halfsina
    return res


def test_praat_interpolate_sinc_matches_the_loop_form_including_the_clipped_depths():
    rng = np.random.Generator(np.random.PCG64(11))
    y = rng.standard_normal(40)
    pos = np.concatenate([rng.uniform(-2.0, 42.0, 300), np.arange(-1, 41, dtype=np.float64), [0.25, 0.75, 1.5, 2.5, 37.5, 38.5, 38.99]])
    for depth in (1, 2, 3, 7, 50):
        got = ro.praat_interpolate_sinc(y, pos, depth)
        ref = np.array([_interpolate_sinc_scalar(y, p + 1.0, depth) for p in pos])
        assert np.abs(got - ref).max() <= 1e-13, depth
    assert np.array_equal(ro.praat_interpolate_sinc(y, np.arange(40.0), 50), y)        # on a sample: that sample


def test_praat_fft_lowpass_is_a_brick_wall_in_praats_packed_order():
    n = 6192                                                                         # nfft = 8192 exactly
    nfft = 8192
    upfactor = 0.5                                                                   # first cleared position 4096 = Im of bin 2047
    # the mask written out per position of NUMrealft's packed array
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.standard_normal(n)
    y = ro.praat_fft_lowpass(x, upfactor)
    data = np.zeros(nfft); data[1000:1000 + n] = x
    packed = np.zeros(nfft)                                                          # NUMrealft order: DC, Nyquist, Re 1, Im 1, ...
    spec = np.fft.rfft(data)
    packed[0], packed[1] = spec[0].real, spec[-1].real
    packed[2::2], packed[3::2] = spec[1:-1].real, spec[1:-1].imag
    packed[int(np.floor(upfactor * nfft)) - 1:] = 0.0                                # 1-based position -> 0-based
    packed[1] = 0.0
    back = np.zeros(nfft // 2 + 1, dtype=np.complex128)
    back[0] = packed[0]
    back[1:-1] = packed[2::2] + 1j * packed[3::2]
    ref = np.fft.irfft(back, nfft)[1000:1000 + n]
    assert np.abs(y - ref).max() <= 1e-12
    assert back[2047].real != 0.0 and back[2047].imag == 0.0                         # the half-cleared bin of an even position
    assert np.abs(y).max() > 0.1 and len(y) == n
    assert ro.praat_fft_lowpass(x[:10], 0.3).shape == (10,)                          # nfft = 2048 for the shortest sounds


def test_sound_resample_removes_what_lies_above_the_new_nyquist():
    fs = 48000.0
    n = 24000
    t = (np.arange(n) + 0.5) / fs
    low, high = np.sin(2 * np.pi * 1234.0 * t), np.sin(2 * np.pi * 9500.0 * t)
    y_both = ro.resample_praat(low + high, fs, 16000.0, 50).astype(np.float64)
    y_low = ro.resample_praat(low, fs, 16000.0, 50).astype(np.float64)
    assert np.abs(y_both - y_low)[300:-300].max() < 2e-3                             # 9.5 kHz is gone, not folded to 6.5 kHz
    same = ro.resample_praat(low, 16000.0, 16000.0, 50)
    assert np.array_equal(same, low.astype(np.float32))


def test_formant_resampler_tables_are_the_full_depth_interpolation_weights():
    """Host logic of the 16 kHz -> 10 kHz resampling inside To Formant (burg): the five polyphase rows reproduce
    NUM_interpolate_sinc wherever Praat does not clip the depth."""
    import math
    from robust_speech_analysis_framework_amd.mshds import DX, RS_RATE, resample10k_tables
    rng = np.random.Generator(np.random.PCG64(3))
    for nn, depth in ((9999, 500), (4800, 50), (3205, 7)):
        y = rng.standard_normal(nn)
        duration = nn * DX
        m = int(math.floor(duration * RS_RATE + 0.5))
        x1o = 0.5 * (duration - (m - 1) / RS_RATE)
        pos0 = (x1o - 0.5 * DX) / DX
        rows, bases = resample10k_tables(pos0, depth)
        ref = ro.praat_interpolate_sinc(y, (x1o + np.arange(m) / RS_RATE - 0.5 * DX) / DX, depth)
        checked = 0
        for mm in range(m):
            q, r = divmod(mm, 5)
            lo = bases[r] + 8 * q - depth
            if lo < 0 or lo + 2 * depth > nn - 1:
                continue
            assert abs(float(np.dot(rows[r], y[lo:lo + 2 * depth + 1])) - ref[mm]) <= 1e-9
            checked += 1
        assert checked > m // 2


def test_resampled_sound_time_axis_known_answers():
    """Sound_resample centres the new grid in the ORIGINAL domain: a tone keeps its phase at t = 0 when read on the
    returned axis.  Sound_upsample (rate ratio 2): even output samples are the input samples (below the 5 % ramp), on a
    grid declared a quarter input period early; the general branch is NOT what a doubling goes through."""
    f0, ph = 440.0, 0.7

    def phase_at_zero(y, x1):
        m = len(y)
        ty = x1 + np.arange(m) / 16000.0
        sl = slice(m // 4, 3 * m // 4)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * ty[sl] / 0.5)
        A = np.stack([np.sin(2 * np.pi * f0 * ty[sl]), np.cos(2 * np.pi * f0 * ty[sl])], 1) * w[:, None]
        c = np.linalg.lstsq(A, y[sl].astype(np.float64), rcond=None)[0]
        return np.arctan2(c[1], c[0])
    for fs in (44100.0, 22050.0):
        n = int(fs * 0.5) + 7
        t = (np.arange(n) + 0.5) / fs
        x = np.sin(2 * np.pi * f0 * t + ph) * (0.5 - 0.5 * np.cos(2 * np.pi * t / 0.5))
        y, x1, xmax = ro.resample_praat_sound(x, fs, 16000.0, 50)
        m = len(y)
        assert m == int(np.floor(n / fs * 16000.0 + 0.5)) and xmax == n / fs
        assert x1 == 0.5 * (xmax - (m - 1) / 16000.0) and abs(x1 - 0.5 / 16000.0) > 1e-7
        assert abs(phase_at_zero(y, x1) - ph) / (2 * np.pi * f0) <= 1e-9
    n = 4000
    t = (np.arange(n) + 0.5) / 8000.0
    x = np.sin(2 * np.pi * f0 * t + ph) * (0.5 - 0.5 * np.cos(2 * np.pi * t / 0.5))
    y, x1, xmax = ro.resample_praat_sound(x, 8000.0, 16000.0, 50)
    assert len(y) == 2 * n and x1 == 0.25 / 8000.0 and xmax == 0.5
    assert np.abs(y[0::2][100:-100] - x[100:-100]).max() <= 1e-5
    # white noise: the ramp over the last 5 % of the packed spectrum removes energy that the general branch (pure sinc
    # interpolation, no filter when the rate goes up) would keep
    rng = np.random.Generator(np.random.PCG64(5))
    wn = rng.standard_normal(3000)
    up = ro.praat_upsample(wn)
    assert len(up) == 6000
    spec = np.abs(np.fft.rfft(up * np.hanning(6000)))
    f = np.fft.rfftfreq(6000, 1.0 / 16000.0)
    assert spec[(f > 4100) & (f < 7900)].max() <= 1e-6 * spec[f < 3000].max()           # nothing above the old Nyquist
    assert spec[(f > 3950) & (f < 4000)].mean() < 0.25 * spec[(f > 3000) & (f < 3700)].mean()   # the ramp
