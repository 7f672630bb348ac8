"""CPU restatement of the two resamplers in front of the extractors.  TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

PARITY UNPINNED: ``torchaudio`` (reference pin 2.5.1) and Praat are absent from this image.
* ``resample_sinc_hann`` restates ``torchaudio.functional.resample`` with the defaults that
  ``torchaudio.transforms.Resample(orig_freq, new_freq)`` uses (``src/foundation_model_extractor.py:93-94``):
  ``sinc_interp_hann``, ``lowpass_filter_width=6``, ``rolloff=0.99``; kernel evaluated in float64 on a float32
  phase grid, stored as float32, applied as a strided correlation on the zero-padded waveform, output cut to
  ``ceil(new * n / orig)`` samples.  Written as a direct double loop over (frame, phase), not as a convolution call.
* ``resample_praat`` restates ``Sound.resample(16000, 50)`` (``src/mshds_extractor.py:419``) with the free choice
  documented in ``mshds_oracle.resample_10k``: Praat's whole-sound FFT low-pass and sinc interpolation are folded
  into one raised-cosine windowed sinc.
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


def resample_praat(x, fs_in: float, fs_out: float = 16000.0, depth: int = 50):
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    if fs_in == fs_out:
        return x.astype(np.float32)
    duration = n / fs_in
    m = int(np.floor(duration * fs_out + 0.5))
    dxi, dxo = 1.0 / fs_in, 1.0 / fs_out
    x1o = 0.5 * (duration - (m - 1) * dxo)
    ratio = min(1.0, fs_out / fs_in)
    out = np.empty(m)
    k = np.arange(-depth, depth + 2)               # every |d| <= depth + 1 for any fractional position
    for i0 in range(0, m, 4096):
        idx = np.arange(i0, min(m, i0 + 4096))
        pos = (x1o + idx * dxo - 0.5 * dxi) / dxi
        base = np.floor(pos).astype(np.int64)
        j = base[:, None] + k[None, :]
        d = pos[:, None] - j
        w = ratio * np.sinc(ratio * d) * (0.5 + 0.5 * np.cos(np.pi * d / (depth + 1.0)))
        w = np.where(np.abs(d) <= depth + 1.0, w, 0.0)
        ok = (j >= 0) & (j < n)
        out[idx] = np.sum(np.where(ok, x[np.clip(j, 0, n - 1)] * w, 0.0), axis=1)
    return out.astype(np.float32)
