"""Deterministic synthetic speech-like clips.

Clip *k* uses ``numpy.random.Generator(PCG64(seed=20260000+k))``; 16 kHz mono;
harmonic source (20 harmonics, -12 dB/oct) with a slowly drifting F0, two fixed
formant resonators, a syllable envelope with pauses, white noise at -35 dB,
peak-normalised to 0.5 and quantised to int16.  Both the CPU oracle and the HIP
path read the same quantised samples (float32 = int16 / 32768), which is what a
PCM-16 WAV loader hands over (reference loaders: ``src/mshds_extractor.py:415``,
``src/foundation_model_extractor.py:87``).
"""
from __future__ import annotations

import os
import wave

import numpy as np
from scipy.signal import lfilter

SAMPLE_RATE = 16000
SEED_BASE = 20260000


def _resonator(fc: float, bw: float, fs: float):
    r = np.exp(-np.pi * bw / fs)
    theta = 2.0 * np.pi * fc / fs
    a = [1.0, -2.0 * r * np.cos(theta), r * r]
    b = [1.0 - r]
    return b, a


def synth_clip_int16(k: int, seconds: float = 30.0, fs: int = SAMPLE_RATE) -> np.ndarray:
    """Return clip *k* as int16 samples (deterministic)."""
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + int(k)))
    n = int(round(seconds * fs))
    t = np.arange(n, dtype=np.float64) / fs

    base = rng.uniform(90.0, 230.0)
    drift_hz = rng.uniform(0.1, 0.5)
    drift_ph = rng.uniform(0.0, 2.0 * np.pi)
    f0 = base * (1.0 + 0.15 * np.sin(2.0 * np.pi * drift_hz * t + drift_ph))
    # 1 % jitter, held per ~10 ms block so it looks like cycle-to-cycle perturbation
    nblk = n // 160 + 1
    jit = 1.0 + 0.01 * rng.standard_normal(nblk)
    f0 = f0 * np.repeat(jit, 160)[:n]
    phase = 2.0 * np.pi * np.cumsum(f0) / fs

    src = np.zeros(n, dtype=np.float64)
    for h in range(1, 21):
        ok = (h * f0) < (0.45 * fs)
        src += np.where(ok, np.sin(h * phase) / (h * h), 0.0)

    y = src
    for fc, bw in ((500.0, 80.0), (1500.0, 120.0)):
        b, a = _resonator(fc, bw, fs)
        y = y + 4.0 * lfilter(b, a, src)

    # syllable envelope with pauses
    env = np.zeros(n, dtype=np.float64)
    pos = rng.uniform(0.05, 0.3)
    next_pause = pos + rng.uniform(1.5, 3.0)
    dur = n / fs
    while pos < dur:
        if pos >= next_pause:
            pos += rng.uniform(0.3, 0.8)
            next_pause = pos + rng.uniform(1.5, 3.0)
            continue
        rate = rng.uniform(3.0, 5.0)
        syl = 1.0 / rate
        i0 = int(pos * fs)
        i1 = min(n, int((pos + syl) * fs))
        if i1 > i0:
            m = i1 - i0
            amp = rng.uniform(0.6, 1.0)
            env[i0:i1] = np.maximum(
                env[i0:i1], amp * 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(m) / m)))
        pos += syl
    y = y * env
    peak = np.max(np.abs(y)) + 1e-12
    y = y / peak
    y = y + (10.0 ** (-35.0 / 20.0)) * rng.standard_normal(n)
    y = 0.5 * y / (np.max(np.abs(y)) + 1e-12)
    return np.round(y * 32767.0).astype(np.int16)


def synth_clip(k: int, seconds: float = 30.0, fs: int = SAMPLE_RATE) -> np.ndarray:
    """Clip *k* as float32 in [-1, 1) (= int16 / 32768, what a PCM-16 WAV reader returns)."""
    return (synth_clip_int16(k, seconds, fs).astype(np.float32) / np.float32(32768.0))


def write_wav(path: str, pcm: np.ndarray, fs: int = SAMPLE_RATE) -> None:
    pcm = np.asarray(pcm)
    if pcm.dtype != np.int16:
        raise TypeError("write_wav expects int16 samples")
    nch = 1 if pcm.ndim == 1 else pcm.shape[1]
    with wave.open(path, "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(2)
        w.setframerate(fs)
        w.writeframes(np.ascontiguousarray(pcm).tobytes())


def write_synth_corpus(directory: str, n_clips: int, seconds: float, first: int = 0):
    """Write clips ``first..first+n_clips-1`` as PCM-16 WAVs; return the file paths."""
    os.makedirs(directory, exist_ok=True)
    paths = []
    for k in range(first, first + n_clips):
        p = os.path.join(directory, f"synth_{k:05d}.wav")
        write_wav(p, synth_clip_int16(k, seconds))
        paths.append(p)
    return paths


def synth_batch(n_clips: int, seconds: float, pool: int | None = None, first: int = 0) -> np.ndarray:
    """[n_clips, n_samples] float32.  With ``pool`` only that many distinct clips are
    synthesised and tiled (bench-sized batches; content does not change the work)."""
    uniq = n_clips if pool is None else min(pool, n_clips)
    base = np.stack([synth_clip(first + k, seconds) for k in range(uniq)])
    if uniq == n_clips:
        return base
    reps = (n_clips + uniq - 1) // uniq
    return np.ascontiguousarray(np.tile(base, (reps, 1))[:n_clips])
