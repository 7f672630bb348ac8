"""CPU restatement of the two resamplers in front of the extractors.  TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

PARITY UNPINNED: ``torchaudio`` (reference pin 2.5.1) and Praat are absent from this image.
* ``resample_sinc_hann`` restates ``torchaudio.functional.resample`` with the defaults that
  ``torchaudio.transforms.Resample(orig_freq, new_freq)`` uses (``src/foundation_model_extractor.py:93-94``):
  ``sinc_interp_hann``, ``lowpass_filter_width=6``, ``rolloff=0.99``; kernel evaluated in float64 on a float32
  phase grid, stored as float32, applied as a strided correlation on the zero-padded waveform, output cut to
  ``ceil(new * n / orig)`` samples.  Written as a direct double loop over (frame, phase), not as a convolution call.
* ``resample_praat`` restates ``Sound.resample(16000, 50)`` (``src/mshds_extractor.py:419``) from Praat's published
  source: whole-sound FFT brick-wall low-pass when the rate goes down (``praat_fft_lowpass``), then
  ``NUM_interpolate_sinc`` of depth 50 on the re-centred sample grid (``praat_interpolate_sinc``); a doubling of the
  rate is ``Sound_upsample`` (``praat_upsample``).  ``resample_praat_sound`` also returns the time axis (x1, xmax) that
  Praat gives the result and that every later analysis depends on.
"""
from __future__ import annotations

import math

import numpy as np


def sinc_hann_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(orig, new)
    o, n = orig // g, new // g
    base_freq = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base_freq)
    kern = np.zeros((n, 2 * width + o), dtype=np.float64)
    for p in range(n):
        ph = float(np.float32(-p) / np.float32(n))
        for k in range(2 * width + o):
            t = (ph + (k - width) / o) * base_freq
            t = min(max(t, -lowpass_filter_width), lowpass_filter_width)
            w = math.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
            t *= math.pi
            s = 1.0 if t == 0.0 else math.sin(t) / t
            kern[p, k] = s * w * (base_freq / o)
    return kern.astype(np.float32), width, o, n


def resample_sinc_hann(x, orig: int, new: int):
    x = np.asarray(x, dtype=np.float32)
    if orig == new:
        return x.copy()
    kern, width, o, n = sinc_hann_kernel(orig, new)
    length = len(x)
    xp = np.concatenate([np.zeros(width, np.float32), x, np.zeros(width + o, np.float32)]).astype(np.float64)
    n_frames = (len(xp) - kern.shape[1]) // o + 1
    out = np.empty((n_frames, n), dtype=np.float64)
    k64 = kern.astype(np.float64)
    for i in range(n_frames):
        out[i] = k64 @ xp[i * o:i * o + kern.shape[1]]
    target = -(-n * length // o)
    return out.reshape(-1)[:target].astype(np.float32)


ANTI_TURN_AROUND = 1000      # zero samples Praat puts on either side of the sound before the FFT low-pass


def praat_fft_lowpass(x, upfactor: float):
    """The anti-aliasing step of Praat's ``Sound_resample`` (published source, ``fon/Sound.cpp``), taken when
    ``upfactor = new_rate * dx < 1``: the sound is copied into a zero buffer of ``nfft`` samples (the first power of two
    that holds it plus 1 000 zeros on either side), transformed by ``NUMrealft``, the packed array is cleared from the
    1-based position ``floor(upfactor * nfft)`` to the end and at position 2 (the Nyquist bin), and transformed back.
    In the packed array position 1 is the DC bin, position 2 the Nyquist bin, positions 2k + 1 / 2k + 2 the real /
    imaginary part of bin k: when the first cleared position is even, the bin it falls in keeps its real part."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    nfft = 1
    while nfft < n + 2 * ANTI_TURN_AROUND:
        nfft *= 2
    data = np.zeros(nfft)
    data[ANTI_TURN_AROUND:ANTI_TURN_AROUND + n] = x
    spec = np.fft.rfft(data)
    first_cleared = int(np.floor(upfactor * nfft))
    k = np.arange(nfft // 2 + 1)
    re = np.where(2 * k + 1 < first_cleared, spec.real, 0.0)
    im = np.where(2 * k + 2 < first_cleared, spec.imag, 0.0)
    re[0] = spec[0].real if first_cleared > 1 else 0.0
    im[0] = 0.0
    re[-1] = im[-1] = 0.0
    return np.fft.irfft(re + 1j * im, nfft)[ANTI_TURN_AROUND:ANTI_TURN_AROUND + n]


def praat_interpolate_sinc(y, pos, depth: int):
    """``NUM_interpolate_sinc`` (published source, ``melder/NUMinterpol.cpp``) at the real 0-based positions ``pos`` of the
    samples ``y``: the depth is cut to the samples that exist on either side; depth 0 -> nearest sample, 1 -> linear,
    2 -> cubic; otherwise a sinc at the rate of ``y`` under a raised cosine that reaches zero one sample beyond the
    outermost sample used on each side (so the window is asymmetric unless the position lies midway)."""
    y = np.asarray(y, dtype=np.float64)
    pos = np.asarray(pos, dtype=np.float64)
    n = len(y)
    out = np.empty(len(pos))
    x = pos + 1.0                                        # Praat's 1-based index
    midleft = np.floor(x).astype(np.int64)
    midright = midleft + 1
    md = np.minimum(np.minimum(depth, midright - 1), n - midleft)
    beyond = x > n
    before = x < 1
    exact = x == midleft
    easy = beyond | before | exact
    out[beyond] = y[n - 1]
    out[before] = y[0]
    sel = exact & ~beyond & ~before
    out[sel] = y[midleft[sel] - 1]
    nearest = ~easy & (md <= 0)
    out[nearest] = y[np.clip(np.floor(x[nearest] + 0.5).astype(np.int64) - 1, 0, n - 1)]
    lin = ~easy & (md == 1)
    yl, yr = y[midleft[lin] - 1], y[np.clip(midright[lin] - 1, 0, n - 1)]
    out[lin] = yl + (x[lin] - midleft[lin]) * (yr - yl)
    cub = ~easy & (md == 2)
    if cub.any():
        ml, mr = midleft[cub], midright[cub]
        yl, yr = y[ml - 1], y[mr - 1]
        dyl = 0.5 * (yr - y[ml - 2])
        dyr = 0.5 * (y[mr] - yl)
        fil, fir = x[cub] - ml, mr - x[cub]
        out[cub] = yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2.0 * (yr - yl)))
    gen = np.nonzero(~easy & (md > 2))[0]
    k = np.arange(depth)
    sgn = np.where(k % 2 == 0, 1.0, -1.0)
    for i0 in range(0, len(gen), 4096):
        g = gen[i0:i0 + 4096]
        xg, ml, mr, d = x[g], midleft[g], midright[g], md[g]
        left, right = mr - d, ml + d
        acc = np.zeros(len(g))
        for a0, span, first, step in ((np.pi * (xg - ml), xg - left + 1.0, ml, -1), (np.pi * (mr - xg), right - xg + 1.0, mr, 1)):
            a = a0[:, None] + np.pi * k[None, :]
            aa = (a0 / span)[:, None] + (np.pi / span)[:, None] * k[None, :]
            w = 0.5 * np.sin(a0)[:, None] * sgn[None, :] / a * (1.0 + np.cos(aa))
            idx = np.clip(first[:, None] + step * k[None, :] - 1, 0, n - 1)
            acc += np.sum(np.where(k[None, :] < d[:, None], y[idx] * w, 0.0), axis=1)
        out[g] = acc
    return out


def praat_upsample(x):
    """Praat's ``Sound_upsample`` (published source, ``fon/Sound.cpp``; ``Sound_resample`` hands it every request whose
    rate ratio is within 1e-6 of 2): the sound goes into a zero buffer of 2 nfft samples behind 1 000 zeros (nfft = the
    first power of two that holds it plus 2 000), ``NUMrealft`` of the first nfft samples, the packed array is scaled by a
    linear ramp (nfft - i) / (nfft - imin) over its 1-based positions i > imin = (integer)(0.95 nfft) and cleared at
    position 2 (the Nyquist bin, which the longer transform would read as ITS Nyquist bin), inverse ``NUMrealft`` of all
    2 nfft positions, output sample i (1-based, 2 n of them) = data[i + 2000] / nfft.  Position 2k + 1 / 2k + 2 hold the
    real / imaginary part of bin k, so the two parts of a bin on the ramp get different factors.  Output sample 2p (0-based)
    is therefore the filtered sound AT input sample p, although the result is declared to start a quarter input period
    before the first input sample (x1 - dx / 4): the labelling is Praat's, and it is kept."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    nfft = 1
    while nfft < n + 2 * ANTI_TURN_AROUND:
        nfft *= 2
    data = np.zeros(nfft)
    data[ANTI_TURN_AROUND:ANTI_TURN_AROUND + n] = x
    spec = np.fft.rfft(data)                              # bins 0 .. nfft / 2
    imin = int(nfft * 0.95)
    k = np.arange(nfft // 2 + 1)

    def ramp(pos):
        return np.where(pos > imin, (nfft - pos) / float(nfft - imin), 1.0)
    re = spec.real * ramp(2 * k + 1)
    im = spec.imag * ramp(2 * k + 2)
    im[0] = 0.0
    re[-1] = im[-1] = 0.0                                 # position 2
    big = np.zeros(nfft + 1, dtype=np.complex128)         # bins 0 .. nfft of the transform of 2 nfft samples
    big[:nfft // 2 + 1] = re + 1j * im
    y = np.fft.irfft(big, 2 * nfft) * 2.0                 # unnormalised inverse / nfft
    return y[2 * ANTI_TURN_AROUND:2 * ANTI_TURN_AROUND + 2 * n]


def sound_resample(x, x1_in: float, dx_in: float, xmin: float, xmax: float, fs_out: float, depth: int):
    """Praat's ``Sound_resample`` of the samples ``x`` (first sample at time ``x1_in``, period ``dx_in``, domain
    ``[xmin, xmax]``): FFT low-pass when the rate goes down, new sample grid centred in the domain, sinc
    interpolation of the given depth; a doubling of the rate is ``Sound_upsample`` (``praat_upsample``), an unchanged rate
    a copy.  Returns (samples float64, x1_out, dx_out)."""
    x = np.asarray(x, dtype=np.float64)
    upfactor = fs_out * dx_in
    if abs(upfactor - 2.0) < 1e-6:
        return praat_upsample(x), x1_in - dx_in / 4.0, dx_in / 2.0
    m = int(np.floor((xmax - xmin) * fs_out + 0.5))
    dxo = 1.0 / fs_out
    x1o = 0.5 * (xmin + xmax - (m - 1) / fs_out)
    if abs(upfactor - 1.0) < 1e-6:
        return x.copy(), x1_in, dx_in
    src = praat_fft_lowpass(x, upfactor) if upfactor < 1.0 else x
    pos = (x1o + np.arange(max(m, 0)) * dxo - x1_in) / dx_in
    return praat_interpolate_sinc(src, pos, depth), x1o, dxo


def resample_praat_sound(x, fs_in: float, fs_out: float = 16000.0, depth: int = 50):
    """``Sound(path).resample(fs_out, depth)`` (``src/mshds_extractor.py:415-419``) -> (samples float32, x1, xmax): a sound
    read from a file has dx = 1 / fs, x1 = 0.5 / fs and the domain [0, n / fs]; the resampled one keeps the domain."""
    x = np.asarray(x, dtype=np.float64)
    dxi = 1.0 / fs_in
    xmax = len(x) / fs_in
    y, x1o, _ = sound_resample(x, 0.5 / fs_in, dxi, 0.0, xmax, fs_out, depth)
    return y.astype(np.float32), x1o, xmax


def resample_praat(x, fs_in: float, fs_out: float = 16000.0, depth: int = 50):
    return resample_praat_sound(x, fs_in, fs_out, depth)[0]
