"""Sample-rate conversion to 16 kHz in front of the extractors.

The reference resamples with two different tools: torchaudio's ``Resample`` in the Wav2Vec2 extractor
(``src/foundation_model_extractor.py:93-94``) and Praat's ``Sound.resample(16000, 50)`` in the MSHDS
extractor (``src/mshds_extractor.py:419``); SMILExtract analyses at the file's own rate (not built: the
openSMILE drop-in still requires 16 kHz input).  Both conversions run on the device through
``librsaf.so``; this module only prepares the polyphase taps and sizes.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib

_TAPS = {}


def sinc_hann_taps(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Non-zero part of torchaudio's ``_get_sinc_resample_kernel`` per output phase.

    Returns (taps float32 [new, K], tap_start int32 [new], orig_reduced, new_reduced): output sample
    ``i*new + p`` is ``sum_k taps[p, k] * x[i*orig + tap_start[p] + k]``.  torchaudio evaluates the kernel in
    float64 on a float32 phase grid and stores it as float32; the taps outside the clamped window are
    (numerically) zero and are dropped."""
    g = math.gcd(int(orig), int(new))
    o, n = int(orig) // g, int(new) // g
    key = (o, n, lowpass_filter_width, rolloff)
    if key in _TAPS:
        return _TAPS[key]
    base_freq = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base_freq)
    idx = np.arange(-width, width + o, dtype=np.float64)[None, :] / o
    phase = (np.arange(0, -n, -1).astype(np.float32) / np.float32(n)).astype(np.float64)[:, None]
    t = (phase + idx) * base_freq
    inside = np.abs(t) < lowpass_filter_width
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
    t = t * math.pi
    scale = base_freq / o
    with np.errstate(divide="ignore", invalid="ignore"):
        kern = np.where(t == 0.0, 1.0, np.sin(t) / t)
    kern = (kern * window * scale).astype(np.float32)
    first = inside.argmax(axis=1)
    last = inside.shape[1] - 1 - inside[:, ::-1].argmax(axis=1)
    K = int((last - first).max()) + 1
    taps = np.zeros((n, K), dtype=np.float32)
    for p in range(n):
        seg = kern[p, first[p]:last[p] + 1]
        taps[p, :len(seg)] = seg
    start = (first - width).astype(np.int32)
    _TAPS[key] = (taps, start, o, n)
    return _TAPS[key]


def resample_sinc_hann(x, orig: int, new: int, device="cuda", stream=None):
    """torchaudio-style resampling of one mono clip: float32 numpy/torch [n] -> torch float32 [ceil(new*n/orig)] on the device."""
    import torch
    lib = _lib.load()
    _lib.require_gpu()
    xd = torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()
    if int(orig) == int(new):
        return xd
    taps, start, o, n = sinc_hann_taps(orig, new)
    n_in = int(xd.numel())
    n_out = -(-n * n_in // o)
    td = torch.from_numpy(taps).to(xd.device)
    sd = torch.from_numpy(start).to(xd.device)
    out = torch.empty(max(n_out, 1), dtype=torch.float32, device=xd.device)
    _lib.check(lib.rsaf_resample_sinc_hann(_lib.ptr(xd), n_in, _lib.ptr(td), _lib.ptr(sd), n, o, taps.shape[1],
                                           _lib.ptr(out), n_out, _lib.stream_ptr(stream)), "rsaf_resample_sinc_hann")
    return out[:n_out]


def praat_resampled_grid(n_in: int, fs_in: float, fs_out: float):
    """Time axis of ``Sound.resample(fs_out, ...)`` applied to a sound of ``n_in`` samples read from a file at ``fs_in`` ->
    (n_out, x1, xmax): Praat keeps the domain [0, xmax = n_in / fs_in], writes round(xmax * fs_out) samples and centres the
    new grid in the domain: x1 = (xmax - (n_out - 1) / fs_out) / 2 (``Sound_resample``; ``Sound_upsample`` for a doubling of
    the rate: 2 n_in samples from x1 - dx / 4, the same numbers).  Unchanged rate: the file's own axis."""
    xmax = n_in / float(fs_in)
    if abs(float(fs_out) * (1.0 / float(fs_in)) - 1.0) < 1e-6:
        return int(n_in), 0.5 / float(fs_in), xmax
    if abs(float(fs_out) * (1.0 / float(fs_in)) - 2.0) < 1e-6:
        return 2 * int(n_in), 0.5 / float(fs_in) - (1.0 / float(fs_in)) / 4.0, xmax
    n_out = int(math.floor(xmax * float(fs_out) + 0.5))
    return n_out, 0.5 * (xmax - (n_out - 1) / float(fs_out)), xmax


def resample_praat_sound(x, fs_in: float, fs_out: float = 16000.0, precision: int = 50, device="cuda", stream=None):
    """``resample_praat`` together with the time axis Praat gives the result -> (samples, x1, xmax); the MSHDS analyses
    take x1 / xmax through ``MshdsEngine.extract_packed(..., x1=, xmax=)``."""
    n_out, x1, xmax = praat_resampled_grid(int(getattr(x, "numel", lambda: len(x))()), fs_in, fs_out)
    return resample_praat(x, fs_in, fs_out, precision, device, stream), x1, xmax


def resample_praat(x, fs_in: float, fs_out: float = 16000.0, precision: int = 50, device="cuda", stream=None):
    """Praat ``Sound.resample(fs_out, precision)`` of one mono clip -> torch float32 [round(n/fs_in*fs_out)] on the device.

    Praat's own steps (``Sound_resample``): whole-sound FFT brick-wall low-pass when the rate goes down, then
    ``NUM_interpolate_sinc`` on the new sample grid centred in the old time domain; a doubling of the rate goes to
    ``Sound_upsample`` instead (spectrum ramped to zero over its last 5 %, inverse transform of twice the length).  The
    samples alone do not say where the grid lies: ``resample_praat_sound`` returns x1 and the domain with them.  Praat
    keeps the resampled sound in float64; it is stored as float32 here like every other clip (one rounding of 6e-8 relative)."""
    import torch
    lib = _lib.load()
    _lib.require_gpu()
    xd = torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()
    if abs(float(fs_out) * (1.0 / float(fs_in)) - 1.0) < 1e-6:           # Sound_resample: a copy
        return xd
    n_in = int(xd.numel())
    n_out = praat_resampled_grid(n_in, fs_in, fs_out)[0]
    out = torch.empty(max(n_out, 1), dtype=torch.float32, device=xd.device)
    wb = int(lib.rsaf_resample_praat_work_bytes(n_in, float(fs_in), float(fs_out)))
    work = torch.empty(max(wb, 8) // 8, dtype=torch.float64, device=xd.device)
    _lib.check(lib.rsaf_resample_praat(_lib.ptr(xd), n_in, float(fs_in), float(fs_out), int(precision), _lib.ptr(out),
                                       n_out, _lib.ptr(work), wb, _lib.stream_ptr(stream)), "rsaf_resample_praat")
    return out[:n_out]
