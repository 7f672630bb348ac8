"""CPU restatement of the openSMILE chain configured by ``Androids.conf``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: openSMILE 3.0.2 is an
external binary the reference spawns (``src/opensmile_extractor.py:62-75``); it is absent
here.  Everything below follows ``Androids.conf`` section by section (line numbers cited)
and the published component descriptions (Eyben et al. 2010, openSMILE book; HTK book
§5 for the HTK-compatible mel/MFCC path).  Where a detail is a free choice it is fixed
here, documented, and mirrored bit-for-bit in ``csrc/smile_lld.hip`` /
``csrc/smile_functionals.hip``.

Arithmetic: float64 on the float32 samples (openSMILE itself is float32).
"""
from __future__ import annotations

import numpy as np

FS = 16000
FRAME = 400          # cFramer frameSize 25 ms   (Androids.conf:73-78)
HOP = 160            # cFramer frameStep 10 ms
NFFT = 512           # cTransformFFT zero-pads to the next power of two (Androids.conf:93-95)
NBINS = NFFT // 2 + 1
DF = FS / NFFT       # 31.25 Hz
PREEMPH = 0.97       # cVectorPreemphasis k (Androids.conf:80-83)
NMEL = 26            # cMelspec nBands default
MEL_LO, MEL_HI = 20.0, 8000.0   # Androids.conf:106-107
NMFCC = 12           # firstMfcc=1..lastMfcc=12 (Androids.conf:112-113)
CEP_LIFTER = 22.0    # cMfcc cepLifter default
MEL_FLOOR = 1.0      # HTK-compatible log floor
HTK_SCALE = 32767.0  # htkcompatible=1 scales samples to the 16-bit range
I0 = 1e-6            # cIntensity reference intensity

# ---- LLD inventory (order = cFunctionals reader order lld;lld2;lld3, Androids.conf:350) ----
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
NLLD = len(LLD_NAMES)            # 38
LEVELS = [(0, 16), (16, 22), (22, 38)]   # lld, lld2, lld3 slices
# LLDs whose kernels are not built yet (SHS pitch + Viterbi, cPitchJitter): columns are NaN
LLD_NOT_BUILT = [14, 15, 18, 19, 20, 21]
FUNCTIONAL_NAMES = ["max", "min", "range", "maxPos", "minPos", "amean",
                    "linregc1", "linregc2", "linregerrQ", "stddev", "skewness", "kurtosis"]
NFUNC = len(FUNCTIONAL_NAMES)    # 12


def feature_names():
    """912 column names in cCsvSink header order (Androids.conf:349-381)."""
    names = []
    for lo, hi in LEVELS:
        for suffix in ("_sma", "_sma_de"):
            for i in range(lo, hi):
                n = LLD_NAMES[i]
                if n.startswith("mfcc["):
                    base = "mfcc" + suffix + n[4:]        # mfcc_sma[1]
                else:
                    base = n + suffix
                names += [f"{base}_{f}" for f in FUNCTIONAL_NAMES]
    return names


def n_frames(n_samples: int) -> int:
    """cFramer: frames only while a full frame exists (integer-exact contract)."""
    return 0 if n_samples < FRAME else (n_samples - FRAME) // HOP + 1


def hamming(n: int = FRAME) -> np.ndarray:
    i = np.arange(n, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * i / (n - 1))


def mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_tables():
    """HTK-style filterbank (HTK book §5.4, FBank): per-bin lower channel and weight.

    Returns (lo_chan[NBINS] int (0..NMEL, -1 = bin unused), lo_wt[NBINS]).
    Channel c (1-based) receives lo_wt*m from bins with lo_chan == c and (1-lo_wt)*m
    from bins with lo_chan == c-1.
    """
    cf = mel(MEL_LO) + (mel(MEL_HI) - mel(MEL_LO)) * np.arange(NMEL + 2) / (NMEL + 1)
    lo_chan = np.full(NBINS, -1, dtype=np.int32)
    lo_wt = np.zeros(NBINS, dtype=np.float64)
    for b in range(NBINS):
        f = b * DF
        if f < MEL_LO or f > MEL_HI:
            continue
        m = float(mel(f))
        c = int(np.searchsorted(cf, m, side="right") - 1)
        c = min(max(c, 0), NMEL)
        lo_chan[b] = c
        lo_wt[b] = (cf[c + 1] - m) / (cf[c + 1] - cf[c])
    return lo_chan, lo_wt


def mel_matrix() -> np.ndarray:
    """Dense [NMEL, NBINS] weight matrix equivalent to ``mel_tables``."""
    lo_chan, lo_wt = mel_tables()
    W = np.zeros((NMEL, NBINS), dtype=np.float64)
    for b in range(NBINS):
        c = lo_chan[b]
        if c < 0:
            continue
        if c >= 1:
            W[c - 1, b] += lo_wt[b]
        if c + 1 <= NMEL:
            W[c, b] += 1.0 - lo_wt[b]
    return W


def dct_matrix() -> np.ndarray:
    """[NMFCC, NMEL] DCT-II rows k=1..12 with HTK scaling and the cepstral lifter folded in."""
    k = np.arange(1, NMFCC + 1, dtype=np.float64)[:, None]
    j = np.arange(1, NMEL + 1, dtype=np.float64)[None, :]
    D = np.sqrt(2.0 / NMEL) * np.cos(np.pi * k * (j - 0.5) / NMEL)
    lift = 1.0 + (CEP_LIFTER / 2.0) * np.sin(np.pi * k / CEP_LIFTER)
    return D * lift


def bark(f):
    f = np.asarray(f, dtype=np.float64)
    return 13.0 * np.arctan(0.00076 * f) + 3.5 * np.arctan((f / 7500.0) ** 2)


def sharpness_weights() -> np.ndarray:
    """Zwicker-style sharpness weighting per bin: bark(f) * g(bark)."""
    z = bark(np.arange(NBINS) * DF)
    g = np.where(z < 14.0, 1.0, 0.066 * np.exp(0.171 * z))
    return z * g


def frame_signal(x: np.ndarray) -> np.ndarray:
    nf = n_frames(len(x))
    if nf == 0:
        return np.zeros((0, FRAME), dtype=np.float64)
    idx = np.arange(FRAME)[None, :] + HOP * np.arange(nf)[:, None]
    return np.asarray(x, dtype=np.float64)[idx]


def lld(x: np.ndarray) -> np.ndarray:
    """38 low-level descriptors per frame for one clip: float64 [NLLD, nF].

    Rows listed in LLD_NOT_BUILT are NaN (kernels not built yet).
    """
    x = np.asarray(x)
    fr = frame_signal(x)                       # raw frames ('frames' level)
    nf = fr.shape[0]
    out = np.full((NLLD, nf), np.nan, dtype=np.float64)
    if nf == 0:
        return out
    # cVectorPreemphasis: per frame, first sample HTK-style (free choice, documented)
    pe = np.empty_like(fr)
    pe[:, 0] = fr[:, 0] * (1.0 - PREEMPH)
    pe[:, 1:] = fr[:, 1:] - PREEMPH * fr[:, :-1]
    ham = hamming()
    win = pe * ham[None, :]                    # 'winframe' level
    spec = np.fft.rfft(win, n=NFFT, axis=1)
    mag = np.abs(spec)                         # 'fftmag' level  [nf, 257]

    # --- cEnergy rms on winframe (Androids.conf:117-123)
    out[0] = np.sqrt(np.sum(win * win, axis=1) / FRAME)
    # --- cMelspec + cMfcc (Androids.conf:101-115)
    melspec = (mag * HTK_SCALE) @ mel_matrix().T
    logmel = np.log(np.maximum(melspec, MEL_FLOOR))
    out[1:13] = (logmel @ dct_matrix().T).T
    # --- cMZcr zcr on raw frames (Androids.conf:125-132): sign changes / N
    out[13] = np.sum(fr[:, 1:] * fr[:, :-1] < 0.0, axis=1) / FRAME
    # --- cIntensity on winframe (Androids.conf:134-139)
    im = np.sum(ham[None, :] * win * win, axis=1) / np.sum(ham)
    out[16] = im / I0
    out[17] = (im / I0) ** 0.3
    # --- cSpectral on fftmag (Androids.conf:258-280), power spectrum (squareInput default)
    P = mag * mag
    f = np.arange(NBINS, dtype=np.float64) * DF
    tot = np.sum(P, axis=1)
    safe = np.where(tot > 0, tot, 1.0)
    out[22] = np.sum(P[:, (f >= 250.0) & (f <= 650.0)], axis=1)
    out[23] = np.sum(P[:, (f >= 1000.0) & (f <= 4000.0)], axis=1)
    cum = np.cumsum(P, axis=1)
    for j, p in enumerate((0.25, 0.50, 0.75, 0.90)):
        # first bin whose inclusive cumulative sum reaches p*total
        idx = np.argmax(cum >= (p * tot)[:, None], axis=1)
        out[24 + j] = idx * DF
    d = np.diff(mag, axis=0, prepend=mag[:1])
    out[28] = np.sqrt(np.sum(d * d, axis=1) / NBINS)          # flux (0 for the first frame)
    cen = np.sum(P * f[None, :], axis=1) / safe
    out[29] = cen
    p = P / safe[:, None]
    out[30] = -np.sum(np.where(p > 0, p * np.log2(np.where(p > 0, p, 1.0)), 0.0), axis=1)
    dev = f[None, :] - cen[:, None]
    var = np.sum(dev ** 2 * p, axis=1)
    out[31] = var
    vs = np.where(var > 0, var, 1.0)
    out[32] = np.sum(dev ** 3 * p, axis=1) / vs ** 1.5
    out[33] = np.sum(dev ** 4 * p, axis=1) / vs ** 2
    sf, sff = np.sum(f), np.sum(f * f)
    out[34] = (NBINS * np.sum(P * f[None, :], axis=1) - sf * tot) / (NBINS * sff - sf * sf)
    out[35] = np.sum(P * sharpness_weights()[None, :], axis=1) / safe
    pk = np.zeros(nf)
    mid = mag[:, 1:-1]
    # excess of every bin over the mean of its two neighbours, clipped at zero: at a spectral peak this is the
    # peak's prominence; unlike a "bin is a strict local maximum" test it is continuous in the magnitudes, so
    # float32 and float64 runs cannot disagree on a decision
    pk = np.sum(np.maximum(mid - 0.5 * (mag[:, :-2] + mag[:, 2:]), 0.0), axis=1)
    msum = np.sum(mag, axis=1)
    out[36] = pk / np.where(msum > 0, msum, 1.0)                 # harmonicity proxy (free choice)
    out[37] = np.exp(np.mean(np.log(np.maximum(P, 1e-30)), axis=1)) / np.maximum(tot / NBINS, 1e-30)
    return out


def sma3(c: np.ndarray) -> np.ndarray:
    """cContourSmoother smaWin=3 (Androids.conf:284-314), edge replication."""
    p = np.pad(c, [(0, 0)] * (c.ndim - 1) + [(1, 1)], mode="edge")
    return (p[..., :-2] + p[..., 1:-1] + p[..., 2:]) / 3.0


def delta2(c: np.ndarray) -> np.ndarray:
    """cDeltaRegression deltawin=2 (Androids.conf:319-347), edge replication."""
    p = np.pad(c, [(0, 0)] * (c.ndim - 1) + [(2, 2)], mode="edge")
    return ((p[..., 3:-1] - p[..., 1:-3]) + 2.0 * (p[..., 4:] - p[..., :-4])) / 10.0


def functionals12(c: np.ndarray) -> np.ndarray:
    """12 functionals of contours c[..., T] -> [..., 12] (Androids.conf:349-368)."""
    T = c.shape[-1]
    t = np.arange(T, dtype=np.float64)
    mx, mn = c.max(axis=-1), c.min(axis=-1)
    amax, amin = c.argmax(axis=-1), c.argmin(axis=-1)       # first occurrence
    mean = c.mean(axis=-1)
    tm = t.mean()
    stt = np.sum((t - tm) ** 2)
    dev = c - mean[..., None]
    m = np.sum(dev * (t - tm), axis=-1) / stt if T > 1 else np.zeros_like(mean)
    b = mean - m * tm
    res = c - (m[..., None] * t + b[..., None])
    errq = np.mean(res * res, axis=-1)
    var = np.mean(dev ** 2, axis=-1)
    sd = np.sqrt(var)
    vs = np.where(var > 0, var, 1.0)
    skew = np.where(var > 0, np.mean(dev ** 3, axis=-1) / vs ** 1.5, 0.0)
    kurt = np.where(var > 0, np.mean(dev ** 4, axis=-1) / vs ** 2, 0.0)
    return np.stack([mx, mn, mx - mn, amax.astype(np.float64), amin.astype(np.float64), mean,
                     m, b, errq, sd, skew, kurt], axis=-1)


def functionals(lld_c: np.ndarray) -> np.ndarray:
    """LLD [38, nF] -> 912 functionals in cCsvSink order."""
    if lld_c.shape[1] == 0:
        return np.full(NLLD * 2 * NFUNC, np.nan)
    s = sma3(lld_c)
    d = delta2(s)
    with np.errstate(invalid="ignore"):
        fs = functionals12(s)
        fd = functionals12(d)
    bad = np.isnan(lld_c).any(axis=1)
    fs[bad] = np.nan
    fd[bad] = np.nan
    parts = []
    for lo, hi in LEVELS:
        parts.append(fs[lo:hi].reshape(-1))
        parts.append(fd[lo:hi].reshape(-1))
    return np.concatenate(parts)


def extract(x: np.ndarray) -> np.ndarray:
    """One clip (float32 samples in [-1,1)) -> 912 features."""
    return functionals(lld(x))
