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

FRAME, HOP, NLLD, NFEAT, NCAND = 400, 160, 38, 912, 6     # FRAME / HOP at 16 kHz
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


_GEOMETRY = {}


def geometry(fs: int = SAMPLE_RATE):
    """(frame, hop, nfft) of the chain at sample rate ``fs`` (``rsaf_smile_geometry``: 25 ms / 10 ms frames,
    Androids.conf:73-78, FFT zero-padded to the next power of two)."""
    fs = int(fs)
    if fs not in _GEOMETRY:
        import ctypes as C
        fr, hp, nf = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().rsaf_smile_geometry(fs, C.byref(fr), C.byref(hp), C.byref(nf)), "rsaf_smile_geometry")
        _GEOMETRY[fs] = (fr.value, hp.value, nf.value)
    return _GEOMETRY[fs]


def n_frames(n_samples: int, fs: int = SAMPLE_RATE) -> int:
    if int(fs) == SAMPLE_RATE:
        return 0 if n_samples < FRAME else (n_samples - FRAME) // HOP + 1
    frame, hop, _ = geometry(fs)
    return 0 if n_samples < frame else (n_samples - frame) // hop + 1


class PackedClips:
    """A batch of mono clips of ONE sample rate concatenated in one device buffer (HBM layout of the path)."""

    def __init__(self, wav, clip_off, frame_off, lengths, frames, fs=SAMPLE_RATE):
        self.wav = wav                  # float32 [total_samples]           (device)
        self.clip_off = clip_off        # int64   [n+1] sample offsets       (device)
        self.frame_off = frame_off      # int64   [n+1] frame offsets        (device)
        self.lengths = lengths          # python list of sample counts
        self.frames = frames            # python list of frame counts
        self.fs = int(fs)
        self.n_clips = len(lengths)
        self.total_samples = int(sum(lengths))
        self.total_frames = int(sum(frames))
        self.max_frames = int(max(frames)) if frames else 0


def pack_clips(clips, device="cuda", fs=SAMPLE_RATE) -> PackedClips:
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
    frames = [n_frames(n, fs) for n in lengths]
    co = np.zeros(len(lengths) + 1, dtype=np.int64)
    co[1:] = np.cumsum(lengths)
    fo = np.zeros(len(lengths) + 1, dtype=np.int64)
    fo[1:] = np.cumsum(frames)
    def small(a):                                          # offsets: pinned staging, queued copy (no host wait)
        t = torch.from_numpy(a)
        return t.pin_memory().to(device, non_blocking=True) if torch.device(device).type == "cuda" else t.to(device)
    return PackedClips(wav, small(co), small(fo), lengths, frames, fs)


def smile_lld(p: PackedClips, stream=None, octave_spectrum=False, return_candidates=False):
    """All 38 LLD contours, float64 [38, total_frames] (contour-major): the frame kernel
    (``rsaf_smile_lld_batch``) followed by the Viterbi smoother / energy gate / jitter pass (``rsaf_smile_pitch_track``).
    ``octave_spectrum`` also returns the cSpecScale level [total_frames, nfft/2+1]; ``return_candidates`` the
    cPitchShs candidates [total_frames, 6, 2] (f0, voicing)."""
    import torch
    lib = _lib.load()
    dev = p.wav.device
    lld = torch.empty((NLLD, max(p.total_frames, 1)), dtype=torch.float64, device=dev)
    cand = torch.zeros((max(p.total_frames, 1), NCAND, 2), dtype=torch.float64, device=dev)
    oct_ = None
    if octave_spectrum:
        oct_ = torch.zeros((max(p.total_frames, 1), geometry(p.fs)[2] // 2 + 1), dtype=torch.float64, device=dev)
    if p.total_frames == 0:
        res = [lld[:, :0]]
    else:
        back = torch.empty((p.total_frames, 8), dtype=torch.uint8, device=dev)
        for c0 in range(0, p.n_clips, 65535):
            n = min(65535, p.n_clips - c0)
            mx = max(p.frames[c0:c0 + n])
            _lib.check(lib.rsaf_smile_lld_batch(
                _lib.ptr(p.wav), _lib.c_void_p_off(p.clip_off, c0), _lib.c_void_p_off(p.frame_off, c0),
                n, mx, p.total_frames, p.fs, _lib.ptr(lld), _lib.ptr(cand),
                _lib.ptr(oct_) if oct_ is not None else None, _lib.stream_ptr(stream)), "rsaf_smile_lld_batch")
            _lib.check(lib.rsaf_smile_pitch_track(
                _lib.ptr(p.wav), _lib.c_void_p_off(p.clip_off, c0), _lib.c_void_p_off(p.frame_off, c0), n,
                p.total_frames, p.fs, _lib.ptr(cand), _lib.ptr(back), _lib.ptr(lld), _lib.stream_ptr(stream)),
                "rsaf_smile_pitch_track")
        res = [lld]
    if octave_spectrum:
        res.append(oct_[:p.total_frames])
    if return_candidates:
        res.append(cand[:p.total_frames])
    return res[0] if len(res) == 1 else tuple(res)


def smile_functionals(lld, p: PackedClips, stream=None, window_frames: int = 0):
    """[n_clips, 912] float64 functionals of the LLD contours (``window_frames`` = 0: whole clip)."""
    import torch
    lib = _lib.load()
    out = torch.empty((p.n_clips, NFEAT), dtype=torch.float64, device=p.wav.device)
    if p.n_clips:
        _lib.check(lib.rsaf_smile_functionals(
            _lib.ptr(lld) if lld.numel() else None, _lib.ptr(p.frame_off), p.n_clips,
            p.total_frames, int(window_frames), _lib.ptr(out), _lib.stream_ptr(stream)), "rsaf_smile_functionals")
    return out


def smile_features(p: PackedClips, stream=None, window_frames: int = 0):
    return smile_functionals(smile_lld(p, stream), p, stream, window_frames)


# ---- Androids.conf validation -------------------------------------------------------------------
_EXPECT = {
    ("fr1", "framesize"): 0.025, ("fr1", "framestep"): 0.010, ("pe2", "k"): 0.97,
    ("mspec", "lofreq"): 20.0, ("mspec", "hifreq"): 8000.0, ("mspec", "htkcompatible"): 1.0,
    ("mspec", "usepower"): 0.0, ("mfcc", "firstmfcc"): 1.0, ("mfcc", "lastmfcc"): 12.0,
    ("delta1", "deltawin"): 2.0, ("delta2", "deltawin"): 2.0, ("delta3", "deltawin"): 2.0,
}


# Options of the pitch chain and of cSpectral that the kernels hard-code (Androids.conf:142-280).  A section that is present must
# carry exactly these values where it sets the option at all (an absent option = the value here, which is also what the kernels
# do); a config that changes one of them is refused instead of being silently analysed with the built-in value.
_EXPECT_IF_PRESENT = {
    "scale": {"scale": "octave", "sourcescale": "lin", "interpmethod": "spline", "minf": 25.0, "maxf": -1.0, "npointstarget": 0.0,
              "specsmooth": 1.0, "specenhance": 1.0, "auditoryweighting": 1.0},
    "shs": {"maxpitch": 620.0, "minpitch": 52.0, "ncandidates": 6.0, "scores": 1.0, "voicing": 1.0, "f0c1": 0.0, "voicingc1": 0.0,
            "f0raw": 1.0, "voicingclip": 1.0, "voicingcutoff": 0.7, "octavecorrection": 0.0, "nharmonics": 15.0,
            "compressionfactor": 0.85, "greedypeakalgo": 1.0},
    "pitchSmooth": {"bufferlength": 30.0, "f0final": 1.0, "f0finalenv": 0.0, "voicingfinalclipped": 0.0, "voicingfinalunclipped": 1.0,
                    "f0raw": 0.0, "voicingc1": 0.0, "voicingclip": 0.0, "wtvv": 10.0, "wtvvd": 5.0, "wtvuv": 10.0, "wthr": 4.0,
                    "wtuu": 0.0, "wlocal": 2.0, "wrange": 1.0},
    "volmerge": {"idx": 0.0, "threshold": 0.001, "removeidx": 1.0, "zerovec": 1.0, "outputval": 0.0},
    "pitchJitter": {"f0field": "F0final", "searchrangerel": 0.25, "jitterlocal": 1.0, "jitterddp": 1.0, "jitterlocalenv": 0.0,
                    "jitterddpenv": 0.0, "shimmerlocal": 1.0, "shimmerlocalenv": 0.0, "onlyvoiced": 0.0, "loghnr": 1.0,
                    "usebrokenjitterthresh": 0.0},
    "spectral": {"bands[0]": "250-650", "bands[1]": "1000-4000", "rolloff[0]": 0.25, "rolloff[1]": 0.5, "rolloff[2]": 0.75,
                 "rolloff[3]": 0.9, "flux": 1.0, "centroid": 1.0, "maxpos": 0.0, "minpos": 0.0, "entropy": 1.0, "variance": 1.0,
                 "skewness": 1.0, "kurtosis": 1.0, "slope": 1.0, "sharpness": 1.0, "tonality": 0.0, "harmonicity": 1.0,
                 "flatness": 1.0},
    "energy": {"rms": 1.0, "log": 0.0},
    "lld": {"smawin": 3.0}, "lld2": {"smawin": 3.0}, "lld3": {"smawin": 3.0},
    "mzcr": {"zcr": 1.0, "amax": 0.0, "mcr": 0.0, "maxmin": 0.0, "dc": 0.0},
    "Int": {"intensity": 1.0, "loudness": 1.0},
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
                cur[k.strip().lower()] = re.split(r"\s+(?:;|//)", v.strip())[0].strip()     # drop a trailing comment
    return sections


def functionals_window(sec, frame_period: float = 0.010) -> int:
    """cFunctionals framing of a parsed config (``Androids.conf:349-356``): 0 = the whole file (frameSize 0 or
    frameMode full), else the window in LLD frames when read literally (frameStep 0 = frameSize: round(frameSize / T))."""
    f = sec.get("functL1", {})
    if f.get("framemode", "").lower().startswith("full"):
        return 0
    size, step = float(f.get("framesize", 0.0)), float(f.get("framestep", 0.0))
    if size < 0 or step < 0:
        raise ValueError("[functL1] negative frameSize / frameStep")
    if size == 0.0:
        return 0
    if step not in (0.0, size):
        raise ValueError("[functL1] overlapping / gapped functional windows (frameStep != frameSize) are not implemented")
    return int(np.floor(size / frame_period + 0.5))


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
    for inst, opts in _EXPECT_IF_PRESENT.items():
        got = sec.get(inst)
        if got is None:
            continue
        for key, want in opts.items():
            if key not in got:
                continue
            val = got[key]
            if isinstance(want, str):
                ok = val.strip().lower() == want.lower()
            else:
                try:
                    ok = abs(float(val) - want) <= 1e-9
                except ValueError:
                    ok = False
            if not ok:
                raise ValueError(f"[{inst}] {key}={val} unsupported (the kernels implement {want})")
    functionals_window(sec)                     # framing must be one the kernels can produce
    return sec


_FRAMING_ANNOUNCED = False


def extract_opensmile_features(input_df, opensmile_exe_path, config_file_path,
                               audio_file_column="filepath", verbose=True, batch_clips=256,
                               functionals="whole-file"):
    """Drop-in for ``src/opensmile_extractor.py:9-103`` backed by the HIP kernels.

    ``opensmile_exe_path`` is accepted for signature compatibility and ignored (no process is
    spawned).  ``config_file_path`` must be the reference's ``Androids.conf`` chain; a missing or
    unsupported config follows the reference's fatal-error convention (message + empty DataFrame,
    ``src/opensmile_extractor.py:41-43``).  Files that cannot be processed are omitted
    (``:89-96``).  Every file is analysed at its own sample rate, as SMILExtract does (frame sizes are
    seconds); files are grouped by rate, one kernel launch per group.

    ``functionals``: how ``[functL1]`` (``Androids.conf:349-356``: ``frameSize=0.025``, ``frameStep=0`` under a
    comment that says "functionals over complete input") is read.  ``"whole-file"`` (default) = one row of
    statistics over the complete input, what the reference's caller assumes (``:80-83`` "the single output row");
    ``"first-window"`` = the literal reading (windows of frameSize, the reference's ``.iloc[0]`` keeps the first).
    No SMILExtract output exists here to decide; see ``oracle/smile_oracle.py``.  PARITY UNPINNED: the numbers
    follow this repository's restatement of the openSMILE components, not a recorded SMILExtract run.
    """
    import pandas as pd
    import torch
    if functionals not in ("whole-file", "first-window"):
        raise ValueError("functionals must be 'whole-file' or 'first-window'")
    try:
        sec = validate_smile_conf(config_file_path)
        window = functionals_window(sec) if functionals == "first-window" else 0
    except (OSError, ValueError) as e:
        print(f"FATAL ERROR: unusable openSMILE config '{config_file_path}': {e}")
        return pd.DataFrame()
    _lib.load()
    _lib.require_gpu()
    global _FRAMING_ANNOUNCED
    if verbose and not _FRAMING_ANNOUNCED:
        # once per process: which reading of [functL1] (Androids.conf:349-356) the numbers follow
        print("[rsaf] openSMILE functionals: " + ("statistics of the FIRST 25 ms window (frameSize=0.025, frameStep=0 read literally)"
                                                   if window else "statistics over the COMPLETE input (the config's own comment; "
                                                   "pass functionals='first-window' for the literal reading of frameSize=0.025 / frameStep=0)"))
        _FRAMING_ANNOUNCED = True
    names = feature_names()
    rows = {}
    paths = list(input_df[audio_file_column])
    for b0 in range(0, len(paths), batch_clips):
        groups = {}                                  # sample rate -> [(position, filename, samples)]
        for k, pth in enumerate(paths[b0:b0 + batch_clips]):
            filename = os.path.basename(pth)
            try:
                x, fs = read_wav_mono(pth)
                if n_frames(len(x), fs) == 0:
                    raise ValueError("shorter than one 25 ms frame")
                groups.setdefault(int(fs), []).append((b0 + k, filename, x))
            except Exception as e:  # per-file failure -> file omitted (reference :89-96)
                if verbose:
                    print(f"ERROR: OpenSMILE failed for file '{filename}'. Stderr: {e}")
        for fs, items in groups.items():
            try:
                p = pack_clips([x for _, _, x in items], fs=fs)
                feats = smile_features(p, window_frames=window)
                torch.cuda.synchronize()
                host = feats.cpu().numpy()
            except _lib.RsafError as e:
                if verbose:
                    for _, filename, _ in items:
                        print(f"ERROR: OpenSMILE failed for file '{filename}'. Stderr: {e}")
                continue
            for (pos, fn, _), r in zip(items, host):
                d = dict(zip(names, r.tolist()))
                d["filename"] = fn
                rows[pos] = d
    if not rows:
        print("Warning: No features were successfully extracted. The returned DataFrame is empty.")
        return pd.DataFrame()
    return pd.DataFrame([rows[k] for k in sorted(rows)])
