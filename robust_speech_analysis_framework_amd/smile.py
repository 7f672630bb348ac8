"""Host side of the openSMILE-style chain (drop-in for ``src/opensmile_extractor.py``).

The reference spawns ``SMILExtract -C Androids.conf`` once per file
(``src/opensmile_extractor.py:62-75``) and reads back a one-row CSV (``:78-87``).  Here a
whole batch of clips is packed into one device buffer and processed by two HIP kernels
(``rsaf_smile_lld_batch`` + ``rsaf_smile_functionals``); the DataFrame contract is unchanged:
feature columns in cCsvSink header order, ``filename`` last, failed files omitted.
"""
from __future__ import annotations

import os
import re

import numpy as np

from . import _lib
from .wavio import read_wav_mono

FRAME, HOP, NLLD, NFEAT = 400, 160, 38, 912
SAMPLE_RATE = 16000

LLD_NAMES = (
    ["pcm_RMSenergy"] + [f"mfcc[{i}]" for i in range(1, 13)] + ["pcm_zcr", "F0final", "voicingFinalUnclipped"]
    + ["pcm_intensity", "pcm_loudness", "jitterLocal", "jitterDDP", "shimmerLocal", "logHNR"]
    + ["pcm_fftMag_fband250-650", "pcm_fftMag_fband1000-4000",
       "pcm_fftMag_spectralRollOff25.0", "pcm_fftMag_spectralRollOff50.0",
       "pcm_fftMag_spectralRollOff75.0", "pcm_fftMag_spectralRollOff90.0",
       "pcm_fftMag_spectralFlux", "pcm_fftMag_spectralCentroid", "pcm_fftMag_spectralEntropy",
       "pcm_fftMag_spectralVariance", "pcm_fftMag_spectralSkewness", "pcm_fftMag_spectralKurtosis",
       "pcm_fftMag_spectralSlope", "pcm_fftMag_psySharpness", "pcm_fftMag_spectralHarmonicity",
       "pcm_fftMag_spectralFlatness"]
)
FUNCTIONAL_NAMES = ["max", "min", "range", "maxPos", "minPos", "amean",
                    "linregc1", "linregc2", "linregerrQ", "stddev", "skewness", "kurtosis"]
_LEVELS = [(0, 16), (16, 22), (22, 38)]


def feature_names():
    """912 column names in cCsvSink header order (Androids.conf:349-381)."""
    names = []
    for lo, hi in _LEVELS:
        for suffix in ("_sma", "_sma_de"):
            for i in range(lo, hi):
                n = LLD_NAMES[i]
                base = ("mfcc" + suffix + n[4:]) if n.startswith("mfcc[") else (n + suffix)
                names += [f"{base}_{f}" for f in FUNCTIONAL_NAMES]
    return names


def n_frames(n_samples: int) -> int:
    return 0 if n_samples < FRAME else (n_samples - FRAME) // HOP + 1


class PackedClips:
    """A batch of mono 16 kHz clips concatenated in one device buffer (HBM layout of the path)."""

    def __init__(self, wav, clip_off, frame_off, lengths, frames):
        self.wav = wav                  # float32 [total_samples]           (device)
        self.clip_off = clip_off        # int64   [n+1] sample offsets       (device)
        self.frame_off = frame_off      # int64   [n+1] frame offsets        (device)
        self.lengths = lengths          # python list of sample counts
        self.frames = frames            # python list of frame counts
        self.n_clips = len(lengths)
        self.total_samples = int(sum(lengths))
        self.total_frames = int(sum(frames))
        self.max_frames = int(max(frames)) if frames else 0


def pack_clips(clips, device="cuda") -> PackedClips:
    """clips: list of 1-D float32 arrays/tensors, or a 2-D [n, samples] array/tensor."""
    import torch
    _lib.require_gpu()
    if isinstance(clips, torch.Tensor) and clips.dim() == 2:
        lengths = [int(clips.shape[1])] * int(clips.shape[0])
        wav = clips.to(device=device, dtype=torch.float32).contiguous().view(-1)
    elif isinstance(clips, np.ndarray) and clips.ndim == 2:
        lengths = [int(clips.shape[1])] * int(clips.shape[0])
        wav = torch.from_numpy(np.ascontiguousarray(clips, dtype=np.float32)).to(device).view(-1)
    else:
        arrs = [np.ascontiguousarray(np.asarray(c, dtype=np.float32).reshape(-1)) for c in clips]
        lengths = [int(a.shape[0]) for a in arrs]
        host = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.float32)
        wav = torch.from_numpy(host).to(device)
    frames = [n_frames(n) for n in lengths]
    co = np.zeros(len(lengths) + 1, dtype=np.int64)
    co[1:] = np.cumsum(lengths)
    fo = np.zeros(len(lengths) + 1, dtype=np.int64)
    fo[1:] = np.cumsum(frames)
    return PackedClips(wav, torch.from_numpy(co).to(device), torch.from_numpy(fo).to(device),
                       lengths, frames)


def smile_lld(p: PackedClips, stream=None):
    """LLD contours, float32 [38, total_frames] (contour-major)."""
    import torch
    lib = _lib.load()
    lld = torch.empty((NLLD, max(p.total_frames, 1)), dtype=torch.float32, device=p.wav.device)
    if p.total_frames == 0:
        return lld[:, :0]
    for c0 in range(0, p.n_clips, 65535):
        n = min(65535, p.n_clips - c0)
        mx = max(p.frames[c0:c0 + n])
        _lib.check(lib.rsaf_smile_lld_batch(
            _lib.ptr(p.wav), _lib.c_void_p_off(p.clip_off, c0), _lib.c_void_p_off(p.frame_off, c0),
            n, mx, p.total_frames, _lib.ptr(lld), _lib.stream_ptr(stream)), "rsaf_smile_lld_batch")
    return lld


def smile_functionals(lld, p: PackedClips, stream=None):
    """[n_clips, 912] float32 functionals of the LLD contours."""
    import torch
    lib = _lib.load()
    out = torch.empty((p.n_clips, NFEAT), dtype=torch.float32, device=p.wav.device)
    if p.n_clips:
        _lib.check(lib.rsaf_smile_functionals(
            _lib.ptr(lld) if lld.numel() else None, _lib.ptr(p.frame_off), p.n_clips,
            p.total_frames, _lib.ptr(out), _lib.stream_ptr(stream)), "rsaf_smile_functionals")
    return out


def smile_features(p: PackedClips, stream=None):
    return smile_functionals(smile_lld(p, stream), p, stream)


# ---- Androids.conf validation -------------------------------------------------------------------
_EXPECT = {
    ("fr1", "framesize"): 0.025, ("fr1", "framestep"): 0.010, ("pe2", "k"): 0.97,
    ("mspec", "lofreq"): 20.0, ("mspec", "hifreq"): 8000.0, ("mspec", "htkcompatible"): 1.0,
    ("mspec", "usepower"): 0.0, ("mfcc", "firstmfcc"): 1.0, ("mfcc", "lastmfcc"): 12.0,
    ("delta1", "deltawin"): 2.0, ("delta2", "deltawin"): 2.0, ("delta3", "deltawin"): 2.0,
}


def parse_smile_conf(path: str):
    """Parse an openSMILE INI-style config into {instance: {key: value}} (lower-cased keys)."""
    sections, cur = {}, None
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        for raw in f:
            line = raw.strip()
            if not line or line.startswith((";", "//", "#")):
                continue
            m = re.match(r"^\[([^:\]]+):([^\]]+)\]", line)
            if m:
                cur = sections.setdefault(m.group(1).strip(), {"__type__": m.group(2).strip()})
                continue
            if cur is not None and "=" in line:
                k, v = line.split("=", 1)
                cur[k.strip().lower()] = v.strip()
    return sections


def validate_smile_conf(path: str):
    """Raise ValueError unless the config is the chain the kernels implement (Androids.conf)."""
    sec = parse_smile_conf(path)
    for (inst, key), want in _EXPECT.items():
        if inst not in sec or key not in sec[inst]:
            raise ValueError(f"config lacks [{inst}] {key}")
        if abs(float(sec[inst][key]) - want) > 1e-9:
            raise ValueError(f"[{inst}] {key}={sec[inst][key]} unsupported (kernels implement {want})")
    if sec.get("w1", {}).get("winfunc", "").lower() != "ham":
        raise ValueError("only the Hamming window is implemented")
    fe = sec.get("functL1", {}).get("functionalsenabled", "")
    if [s.strip() for s in fe.split(";")] != ["Extremes", "Regression", "Moments"]:
        raise ValueError("functionalsEnabled must be Extremes;Regression;Moments")
    return sec


def extract_opensmile_features(input_df, opensmile_exe_path, config_file_path,
                               audio_file_column="filepath", verbose=True, batch_clips=256):
    """Drop-in for ``src/opensmile_extractor.py:9-103`` backed by the HIP kernels.

    ``opensmile_exe_path`` is accepted for signature compatibility and ignored (no process is
    spawned).  ``config_file_path`` must be the reference's ``Androids.conf`` chain; a missing or
    unsupported config follows the reference's fatal-error convention (message + empty DataFrame,
    ``src/opensmile_extractor.py:41-43``).  Files that cannot be processed are omitted
    (``:89-96``).  The six LLDs whose kernels are not built yet give NaN columns.
    """
    import pandas as pd
    import torch
    try:
        validate_smile_conf(config_file_path)
    except (OSError, ValueError) as e:
        print(f"FATAL ERROR: unusable openSMILE config '{config_file_path}': {e}")
        return pd.DataFrame()
    _lib.load()
    _lib.require_gpu()
    names = feature_names()
    rows = []
    paths = list(input_df[audio_file_column])
    for b0 in range(0, len(paths), batch_clips):
        clips, fnames = [], []
        for pth in paths[b0:b0 + batch_clips]:
            filename = os.path.basename(pth)
            try:
                x, fs = read_wav_mono(pth)
                if fs != SAMPLE_RATE:
                    # SMILExtract analyses at the file's own rate (frame sizes in seconds); that is not built.
                    # Opt-in approximation: convert to 16 kHz on the device and run the 16 kHz chain.
                    if os.environ.get("RSAF_SMILE_RESAMPLE", "0") != "1":
                        raise ValueError(f"sample rate {fs} Hz: only 16 kHz input is supported "
                                         "(set RSAF_SMILE_RESAMPLE=1 to convert to 16 kHz first; results then "
                                         "differ from a native-rate analysis)")
                    from .resample import resample_sinc_hann
                    x = resample_sinc_hann(x, fs, SAMPLE_RATE).cpu().numpy()
                if n_frames(len(x)) == 0:
                    raise ValueError("shorter than one 25 ms frame")
                clips.append(x)
                fnames.append(filename)
            except Exception as e:  # per-file failure -> file omitted (reference :89-96)
                if verbose:
                    print(f"ERROR: OpenSMILE failed for file '{filename}'. Stderr: {e}")
        if not clips:
            continue
        p = pack_clips(clips)
        feats = smile_features(p)
        torch.cuda.synchronize()
        host = feats.cpu().numpy()
        for fn, r in zip(fnames, host):
            d = dict(zip(names, r.tolist()))
            d["filename"] = fn
            rows.append(d)
    if not rows:
        print("Warning: No features were successfully extracted. The returned DataFrame is empty.")
        return pd.DataFrame()
    return pd.DataFrame(rows)
