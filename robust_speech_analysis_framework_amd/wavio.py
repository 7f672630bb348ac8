"""Minimal WAV reader for the drop-in extractors.

The reference loads audio through Praat (``parselmouth.Sound``,
``src/mshds_extractor.py:415``), torchaudio (``src/foundation_model_extractor.py:87``)
or SMILExtract's cWaveSource (``Androids.conf:67-71``).  None of those exist in this
image, so the drop-ins read PCM WAV with the standard library and hand float32
samples in [-1, 1) plus the sample rate to the device path.  Mono mix-down is the
channel mean, as in all three reference loaders.
"""
from __future__ import annotations

import wave

import numpy as np


class UnsupportedAudio(ValueError):
    pass


def read_wav(path: str):
    """Return (samples float32 [n_channels, n], sample_rate)."""
    with wave.open(path, "rb") as w:
        nch, width, fs, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / np.float32(32768.0)
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / np.float32(128.0)
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
        x = x.astype(np.float32)
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= (1 << 23), v - (1 << 24), v)
        x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    else:
        raise UnsupportedAudio(f"{path}: unsupported sample width {width}")
    x = x.reshape(-1, nch).T
    return np.ascontiguousarray(x), int(fs)


def read_wav_mono(path: str):
    """Return (mono float32 [n], sample_rate); channel mean as in the reference loaders."""
    x, fs = read_wav(path)
    if x.shape[0] > 1:
        x = x.mean(axis=0, dtype=np.float32)
    else:
        x = x[0]
    return np.ascontiguousarray(x, dtype=np.float32), fs


def read_wav_raw(path: str):
    """Return (raw interleaved PCM bytes, n_channels, sample_width, sample_rate, n_frames) without decoding."""
    with wave.open(path, "rb") as w:
        nch, width, fs, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width not in (1, 2, 3, 4):
        raise UnsupportedAudio(f"{path}: unsupported sample width {width}")
    n = len(raw) // (nch * width)
    return raw, int(nch), int(width), int(fs), int(n)


def read_wav_mono_device(path: str, device="cuda", stream=None):
    """PCM decode and mono mix-down on the device: (torch float32 [n] on ``device``, sample_rate, n_frames).
    Bit-identical to ``read_wav_mono`` (same float32 operations in the same order)."""
    import torch
    from . import _lib
    lib = _lib.load()
    _lib.require_gpu()
    raw, nch, width, fs, n = read_wav_raw(path)
    out = torch.empty(max(n, 1), dtype=torch.float32, device=device)
    if n:
        buf = torch.frombuffer(bytearray(raw[:n * nch * width]), dtype=torch.uint8).to(out.device)
        _lib.check(lib.rsaf_pcm_to_mono_f32(_lib.ptr(buf), width, nch, n, _lib.ptr(out), _lib.stream_ptr(stream)),
                   "rsaf_pcm_to_mono_f32")
    return out[:n], fs, n
