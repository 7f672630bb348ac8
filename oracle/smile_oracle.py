"""CPU restatement of the openSMILE chain configured by ``Androids.conf``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: openSMILE 3.0.2 is an
external binary the reference spawns (``src/opensmile_extractor.py:62-75``); it is absent
here and the reference holds no recorded output of it.  Everything below follows
``Androids.conf`` section by section (line numbers cited) and the published component
descriptions (Eyben et al. 2010, openSMILE book; HTK book §5 for the HTK-compatible mel/MFCC path;
Hermes 1988 and the Praat manual's "Sound: To Pitch (shs)" for the sub-harmonic summation that
cSpecScale + cPitchShs implement).  Where a detail is a free choice it is fixed here, documented
(search for "free choice"), and mirrored in ``csrc/smile_*.hip``.

Arithmetic: float64 on the float32 samples (openSMILE itself is float32).  The HIP chain is float64 as well since round 3
(``csrc/smile_lld.hip``): every decision of the chain then coincides with this restatement's.

All 38 low-level descriptors are built.  The analysis runs at the FILE'S OWN sample rate
(``Params(fs)``): cFramer's frameSize / frameStep are seconds (``Androids.conf:73-78``) and the
reference hands the file to SMILExtract as it is (``src/opensmile_extractor.py:62-69``); the
``sampleRate = 44100`` line of cWaveSource (``:70``) only applies to header-less input.

cFunctionals framing (``Androids.conf:349-356``) -- the one free choice that touches every column:
the ACTIVE lines are ``frameSize=0.025`` / ``frameStep=0`` while the comment above them says
"frameSize and frameStep = 0 => functionals over complete input" and the ``frameSize=0`` form is
commented out.  Two readings exist:
  (a) whole-file functionals: what the comment describes, what the reference's caller assumes
      ("Parse the features from the single output row", ``src/opensmile_extractor.py:80-83``) and
      what the IS09 config this file was derived from does;
  (b) the literal one: openSMILE documents ``frameStep = 0`` as "same as frameSize", which would make
      cFunctionals emit one row per 25 ms window (= round(0.025 / 0.01) = 3 LLD frames), and the
      reference's ``.iloc[0]`` (``:83``) would keep the row of the first window only.
No SMILExtract output exists to decide.  ADOPTED: (a), ``window_frames = 0``; (b) is available to every
function here (``window_frames = 3``) and to the drop-in (``functionals="first-window"``), which
computes the same 12 statistics over the first 3 frames of the full-length sma / delta contours.
"""
from __future__ import annotations

import math

import numpy as np

PREEMPH = 0.97       # cVectorPreemphasis k (Androids.conf:80-83)
NMEL = 26            # cMelspec nBands default
MEL_LO, MEL_HI = 20.0, 8000.0   # Androids.conf:106-107
NMFCC = 12           # firstMfcc=1..lastMfcc=12 (Androids.conf:112-113)
CEP_LIFTER = 22.0    # cMfcc cepLifter default
MEL_FLOOR = 1.0      # HTK-compatible log floor
HTK_SCALE = 32767.0  # htkcompatible=1 scales samples to the 16-bit range
I0 = 1e-6            # cIntensity reference intensity

# cSpecScale (Androids.conf:142-160)
SCALE_MINF = 25.0
# cPitchShs (:162-186)
SHS_NHARM = 15
SHS_COMPRESSION = 0.85
SHS_NCAND = 6
SHS_MINPITCH, SHS_MAXPITCH = 52.0, 620.0
VOICING_CUTOFF = 0.7
# cPitchSmootherViterbi (:190-214)
VIT_BUFLEN = 30
VIT_WTVV, VIT_WTVVD, VIT_WTVUV, VIT_WTHR, VIT_WTUU, VIT_WLOCAL, VIT_WRANGE = 10.0, 5.0, 10.0, 4.0, 0.0, 2.0, 1.0
# cValbasedSelector (:217-229)
ENERGY_GATE = 0.001
# cPitchJitter (:233-255)
JIT_SEARCH_REL = 0.25
JIT_CC_MIN, JIT_CC_MAX = 1e-3, 1.0 - 1e-6

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
I_F0, I_VOICE, I_JL, I_JD, I_SH, I_HNR = 14, 15, 18, 19, 20, 21
FUNCTIONAL_NAMES = ["max", "min", "range", "maxPos", "minPos", "amean",
                    "linregc1", "linregc2", "linregerrQ", "stddev", "skewness", "kurtosis"]
NFUNC = len(FUNCTIONAL_NAMES)    # 12


def _round_half_up(x: float) -> int:
    return int(math.floor(x + 0.5))


class Params:
    """Frame geometry of the chain at sample rate ``fs`` (cFramer sizes are seconds, Androids.conf:73-78).

    frame = round(0.025 / T), hop = round(0.010 / T) with T = 1 / fs in double and round = floor(x + 0.5)
    (free choice: 44.1 kHz -> 1103 / 441, 22.05 kHz -> 551 / 221); cTransformFFT zero-pads the frame to the
    next power of two (Androids.conf:93-95)."""

    def __init__(self, fs: int = 16000):
        self.fs = int(fs)
        T = 1.0 / self.fs
        self.frame = _round_half_up(0.025 / T)
        self.hop = _round_half_up(0.010 / T)
        self.nfft = 1 << max(1, (self.frame - 1).bit_length())
        self.nbins = self.nfft // 2 + 1
        self.df = self.fs / self.nfft
        # cSpecScale: octave scale from 25 Hz to the source's top frequency, as many points as source bins
        # (nPointsTarget = 0, maxF = -1, Androids.conf:155-157)
        self.npts = self.nbins
        self.fmin_l2 = math.log2(SCALE_MINF)
        self.fmax_l2 = math.log2(self.fs / 2.0)
        self.dl2 = (self.fmax_l2 - self.fmin_l2) / (self.npts - 1)       # octaves per target point
        self.ppo = 1.0 / self.dl2                                        # points per octave (real)

    def n_frames(self, n_samples: int) -> int:
        return 0 if n_samples < self.frame else (n_samples - self.frame) // self.hop + 1

    # ---- tables ----
    def hamming(self) -> np.ndarray:
        i = np.arange(self.frame, dtype=np.float64)
        return 0.54 - 0.46 * np.cos(2.0 * np.pi * i / (self.frame - 1))

    def mel_tables(self):
        """HTK-style filterbank (HTK book §5.4, FBank): per-bin lower channel and weight.

        Returns (lo_chan[nbins] int (0..NMEL, -1 = bin unused), lo_wt[nbins]).  Channel c (1-based) receives
        lo_wt*m from bins with lo_chan == c and (1-lo_wt)*m from bins with lo_chan == c-1.  hifreq is clipped
        to the Nyquist frequency (free choice; matters below 16 kHz only)."""
        hi = min(MEL_HI, self.fs / 2.0)
        cf = mel(MEL_LO) + (mel(hi) - mel(MEL_LO)) * np.arange(NMEL + 2) / (NMEL + 1)
        lo_chan = np.full(self.nbins, -1, dtype=np.int32)
        lo_wt = np.zeros(self.nbins, dtype=np.float64)
        for b in range(self.nbins):
            f = b * self.df
            if f < MEL_LO or f > hi:
                continue
            m = float(mel(f))
            c = int(np.searchsorted(cf, m, side="right") - 1)
            c = min(max(c, 0), NMEL)
            lo_chan[b] = c
            lo_wt[b] = (cf[c + 1] - m) / (cf[c + 1] - cf[c])
        return lo_chan, lo_wt

    def mel_matrix(self) -> np.ndarray:
        lo_chan, lo_wt = self.mel_tables()
        W = np.zeros((NMEL, self.nbins), dtype=np.float64)
        for b in range(self.nbins):
            c = lo_chan[b]
            if c < 0:
                continue
            if c >= 1:
                W[c - 1, b] += lo_wt[b]
            if c + 1 <= NMEL:
                W[c, b] += 1.0 - lo_wt[b]
        return W

    def sharpness_weights(self) -> np.ndarray:
        z = bark(np.arange(self.nbins) * self.df)
        g = np.where(z < 14.0, 1.0, 0.066 * np.exp(0.171 * z))
        return z * g

    # ---- cSpecScale / cPitchShs tables ----
    def target_pos(self) -> np.ndarray:
        """Position (in source bins, real) of every octave-scale target point."""
        return np.exp2(self.fmin_l2 + self.dl2 * np.arange(self.npts)) / self.df

    def auditory_weights(self) -> np.ndarray:
        """Hermes' raised arctangent as in Praat's Sound_to_Pitch_shs: 0.5 + atan(3 (i + 1 - atans) / ppo) / pi,
        atans = ppo log2(65 / 50) - 1, i = 0-based target index."""
        atans = self.ppo * math.log2(65.0 / 50.0) - 1.0
        i = np.arange(self.npts, dtype=np.float64)
        return 0.5 + np.arctan(3.0 * (i + 1.0 - atans) / self.ppo) / np.pi

    def shs_shifts(self) -> np.ndarray:
        """floor(ppo * log2(h)) for h = 1..15 (compression weight 0.85^(h-1))."""
        return np.array([int(math.floor(self.ppo * math.log2(h))) for h in range(1, SHS_NHARM + 1)], dtype=np.int64)


_P16 = Params(16000)
FS = 16000
FRAME, HOP, NFFT, NBINS, DF = _P16.frame, _P16.hop, _P16.nfft, _P16.nbins, _P16.df


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


def n_frames(n_samples: int, P: Params = _P16) -> int:
    """cFramer: frames only while a full frame exists (integer-exact contract)."""
    return P.n_frames(n_samples)


def hamming(n: int = FRAME) -> np.ndarray:
    i = np.arange(n, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * i / (n - 1))


def mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_tables(P: Params = _P16):
    return P.mel_tables()


def mel_matrix(P: Params = _P16) -> np.ndarray:
    """Dense [NMEL, nbins] weight matrix equivalent to ``mel_tables``."""
    return P.mel_matrix()


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


def sharpness_weights(P: Params = _P16) -> np.ndarray:
    """Zwicker-style sharpness weighting per bin: bark(f) * g(bark)."""
    return P.sharpness_weights()


def frame_signal(x: np.ndarray, P: Params = _P16) -> np.ndarray:
    nf = P.n_frames(len(x))
    if nf == 0:
        return np.zeros((0, P.frame), dtype=np.float64)
    idx = np.arange(P.frame)[None, :] + P.hop * np.arange(nf)[:, None]
    return np.asarray(x, dtype=np.float64)[idx]


def magnitudes(x: np.ndarray, P: Params = _P16):
    """('frames', 'winframe', 'fftmag') levels of one clip: raw frames, pre-emphasised + Hamming frames, |FFT|."""
    fr = frame_signal(np.asarray(x), P)
    # cVectorPreemphasis: per frame, first sample HTK-style (free choice, documented)
    pe = np.empty_like(fr)
    if fr.shape[0]:
        pe[:, 0] = fr[:, 0] * (1.0 - PREEMPH)
        pe[:, 1:] = fr[:, 1:] - PREEMPH * fr[:, :-1]
    win = pe * P.hamming()[None, :]
    mag = np.abs(np.fft.rfft(win, n=P.nfft, axis=1)) if fr.shape[0] else np.zeros((0, P.nbins))
    return fr, win, mag


# =====================================================================================================
# cSpecScale -> cPitchShs (Androids.conf:142-186): sub-harmonic summation pitch candidates
# =====================================================================================================
def spec_enhance(a: np.ndarray) -> np.ndarray:
    """Hermes' peak enhancement as in Praat's spec_enhance_SHS: keep every bin within 2 bins of a local maximum
    (a[i] > a[i-1] and a[i] >= a[i+1]; the ends count when larger than their one neighbour), zero the rest of
    every stretch BETWEEN two maxima; with a single maximum everything further than 2 bins from it is zeroed."""
    a = np.array(a, dtype=np.float64)
    n = len(a)
    if n < 2:
        return a
    ismax = np.zeros(n, dtype=bool)
    ismax[0] = a[0] > a[1]
    ismax[1:-1] = (a[1:-1] > a[:-2]) & (a[1:-1] >= a[2:])
    ismax[-1] = a[-1] > a[-2]
    pos = np.flatnonzero(ismax)
    if len(pos) == 0:
        return a
    j = np.arange(n)
    near = np.zeros(n, dtype=bool)
    for d in range(-2, 3):
        sh = pos + d
        near[sh[(sh >= 0) & (sh < n)]] = True
    if len(pos) == 1:
        a[~near] = 0.0
    else:
        a[~near & (j > pos[0]) & (j < pos[-1])] = 0.0
    return a


def spec_smooth(a: np.ndarray) -> np.ndarray:
    """Praat's spec_smooth_SHS: (1, 2, 1) / 4 with a zero left of the first bin; the last bin is left alone."""
    a = np.asarray(a, dtype=np.float64)
    out = a.copy()
    if len(a) >= 2:
        left = np.concatenate([[0.0], a[:-2]])
        out[:-1] = (left + 2.0 * a[:-1] + a[1:]) / 4.0
    return out


def natural_spline_m(y: np.ndarray) -> np.ndarray:
    """m_i = y''_i h^2 / 6 of the natural cubic spline through equally spaced knots:
    m_{i-1} + 4 m_i + m_{i+1} = y_{i-1} - 2 y_i + y_{i+1}, m_0 = m_{n-1} = 0 (Thomas algorithm)."""
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    m = np.zeros(n)
    if n < 3:
        return m
    r = y[:-2] - 2.0 * y[1:-1] + y[2:]
    k = n - 2
    cp = np.zeros(k)
    dp = np.zeros(k)
    cp[0] = 1.0 / 4.0
    dp[0] = r[0] / 4.0
    for i in range(1, k):
        den = 4.0 - cp[i - 1]
        cp[i] = 1.0 / den
        dp[i] = (r[i] - dp[i - 1]) / den
    sol = np.zeros(k)
    sol[-1] = dp[-1]
    for i in range(k - 2, -1, -1):
        sol[i] = dp[i] - cp[i] * sol[i + 1]
    m[1:-1] = sol
    return m


def spec_scale(mag: np.ndarray, P: Params = _P16) -> np.ndarray:
    """cSpecScale (Androids.conf:142-160): one magnitude spectrum -> octave-scale spectrum of npts points from 25 Hz
    to fs/2: peak enhancement, smoothing (both on the linear spectrum, "before the scale transformation"), natural
    cubic spline through the equally spaced source bins evaluated at the target frequencies (free choice: spline
    abscissa = linear frequency, sourceScale = lin), negative values reset to 0, auditory weighting."""
    a = spec_smooth(spec_enhance(mag))
    m = natural_spline_m(a)
    pos = P.target_pos()
    klo = np.minimum(np.floor(pos).astype(np.int64), P.nbins - 2)
    b = pos - klo
    aa = 1.0 - b
    y = aa * a[klo] + b * a[klo + 1] + (aa ** 3 - aa) * m[klo] + (b ** 3 - b) * m[klo + 1]
    return np.maximum(y, 0.0) * P.auditory_weights()


def shs_sum(S: np.ndarray, P: Params = _P16) -> np.ndarray:
    """cPitchShs: sum of the octave spectrum shifted by floor(ppo log2 h), h = 1..15, weights 0.85^(h-1)."""
    n = len(S)
    out = np.zeros(n)
    for h, sh in enumerate(P.shs_shifts()):
        if sh < n:
            out[:n - sh] += (SHS_COMPRESSION ** h) * S[sh:]
    return out


def shs_candidates(S: np.ndarray, P: Params = _P16):
    """Up to 6 (f0, voicing, score) triples of one frame, best score first (greedyPeakAlgo = 1: the highest peaks
    regardless of their order; ties: lower frequency first).  Peak = local maximum (y2 > y1 and y2 >= y3) of the
    summation spectrum with parabolic refinement, kept when 52 <= f0 <= 620 Hz; voicing = 1 - mean(SHS) / score
    (clipped at 0).  Missing slots are (0, 0, 0)."""
    H = shs_sum(S, P)
    mean = H.mean()
    i = np.arange(1, len(H) - 1)
    y1, y2, y3 = H[:-2], H[1:-1], H[2:]
    pk = (y2 > y1) & (y2 >= y3)
    den = y1 - 2.0 * y2 + y3
    den = np.where(pk, den, -1.0)
    dx = 0.5 * (y1 - y3) / den
    score = y2 - 0.125 * (y1 - y3) ** 2 / den
    f = np.exp2(P.fmin_l2 + (i + dx) * P.dl2)
    ok = pk & (f >= SHS_MINPITCH) & (f <= SHS_MAXPITCH) & (score > 0.0)
    idx = np.flatnonzero(ok)
    order = idx[np.argsort(-score[idx], kind="stable")][:SHS_NCAND]
    out = np.zeros((SHS_NCAND, 3))
    for s, k in enumerate(order):
        out[s] = (f[k], max(0.0, 1.0 - mean / score[k]), score[k])
    return out


# =====================================================================================================
# cPitchSmootherViterbi (Androids.conf:190-214)
# =====================================================================================================
def viterbi_smooth(cands: np.ndarray):
    """cands [T, 6, 3] (f0, voicing, score; f0 = 0 marks an empty slot) -> (F0final [T], voicingFinalUnclipped [T]).

    States per frame: the 6 candidate slots + "unvoiced".  Free choices (the component's source is not available;
    the weights and their meaning are the config's / the component help's):
      local cost   voiced k : wLocal * -ln(max(v_k, 1e-3)) + (wThr if v_k < voicingCutoff)
                   unvoiced : wLocal * -ln(max(1 - v_best, 1e-3)) + (wThr if v_best >= voicingCutoff),
                   v_best = highest voicing among the frame's candidates (0 without candidates);
                   wRange penalises candidates outside [minPitch, maxPitch]: none exist (range-limited at picking);
      transition   voiced i -> voiced j : wTvv |d| + wTvvd |d - d_i|, d = log2(f_j / f_i), d_i = the d of the best
                   path into i (0 after an unvoiced frame or at the start);
                   voiced <-> unvoiced : wTvuv;  unvoiced -> unvoiced : wTuu;
      decoding     fixed lag: the decision for frame t is read off the best path ending at frame
                   min(t + bufferLength - 1, T - 1) (output index = input index: every level keeps the frame period).
    voicingFinalUnclipped = voicing of the chosen candidate, or v_best when the frame is decided unvoiced."""
    T = cands.shape[0]
    K = SHS_NCAND
    U = K
    f0, vo = cands[:, :, 0], cands[:, :, 1]
    have = f0 > 0.0
    vbest = np.where(have, vo, 0.0).max(axis=1) if T else np.zeros(0)
    INF = 1e30
    local = np.full((T, K + 1), INF)
    lv = VIT_WLOCAL * -np.log(np.maximum(vo, 1e-3)) + np.where(vo < VOICING_CUTOFF, VIT_WTHR, 0.0)
    local[:, :K] = np.where(have, lv, INF)
    local[:, U] = VIT_WLOCAL * -np.log(np.maximum(1.0 - vbest, 1e-3)) + np.where(vbest >= VOICING_CUTOFF, VIT_WTHR, 0.0)
    l2f = np.log2(np.where(have, f0, 1.0))
    cost = np.zeros((T, K + 1))
    back = np.zeros((T, K + 1), dtype=np.int64)
    slope = np.zeros((T, K + 1))
    if T == 0:
        return np.zeros(0), np.zeros(0)
    cost[0] = local[0]
    best_end = np.zeros(T, dtype=np.int64)
    best_end[0] = int(np.argmin(cost[0]))
    for t in range(1, T):
        prev = cost[t - 1] - cost[t - 1].min()               # renormalised: only differences matter
        for j in range(K + 1):
            if local[t, j] >= INF:
                cost[t, j] = INF
                continue
            bc, bi, bd = INF, 0, 0.0
            for i in range(K + 1):
                if prev[i] >= INF:
                    continue
                if i == U and j == U:
                    tr, d = VIT_WTUU, 0.0
                elif i == U or j == U:
                    tr, d = VIT_WTVUV, 0.0
                else:
                    d = l2f[t, j] - l2f[t - 1, i]
                    tr = VIT_WTVV * abs(d) + VIT_WTVVD * abs(d - slope[t - 1, i])
                c = prev[i] + tr
                if c < bc:                                     # ties: lowest predecessor index
                    bc, bi, bd = c, i, d
            cost[t, j] = bc + local[t, j]
            back[t, j] = bi
            slope[t, j] = bd
        best_end[t] = int(np.argmin(cost[t]))
    F = np.zeros(T)
    V = np.zeros(T)
    for t in range(T):
        e = min(t + VIT_BUFLEN - 1, T - 1)
        s = best_end[e]
        for u in range(e, t, -1):
            s = back[u, s]
        if s == U:
            F[t], V[t] = 0.0, vbest[t]
        else:
            F[t], V[t] = f0[t, s], vo[t, s]
    return F, V


def energy_gate(F: np.ndarray, V: np.ndarray, rms: np.ndarray):
    """cValbasedSelector (Androids.conf:217-229): zero the pitch vector where pcm_RMSenergy < 0.001."""
    keep = rms >= ENERGY_GATE
    return np.where(keep, F, 0.0), np.where(keep, V, 0.0)


# =====================================================================================================
# cPitchJitter (Androids.conf:233-255)
# =====================================================================================================
def jitter_shimmer(x: np.ndarray, F0: np.ndarray, P: Params = _P16):
    """Waveform-matched pitch periods on the raw samples, driven by F0final -> (jitterLocal, jitterDDP,
    shimmerLocal, logHNR) per frame.  Free choices (component source unavailable; options from the config):
      * frame t owns the sample span [t hop, (t + 1) hop).  In a run of voiced frames (F0final > 0) a chain of periods
        starts at the first sample of the run's first frame; a period that STARTS at integer sample p (p = floor of the
        chain position) inside the span of frame t is searched with that frame's F0: nominal length T0 = fs / F0, lags
        tau in [ceil(0.75 T0), floor(1.25 T0)] (searchRangeRel = 0.25), normalised cross-correlation over
        W = round(T0) samples  cc(tau) = sum x[p+n] x[p+n+tau] / sqrt(sum x[p+n]^2 * sum x[p+n+tau]^2)  (0 when a sum of
        squares is 0); best tau = first maximum; three-point parabolic refinement when both neighbours are inside
        the lag range -> period length Tp, peak value cc*; amplitude = max - min over x[p .. p+tau_best-1]; the chain
        advances by Tp.  The chain stops for good when p + W + floor(1.25 T0) would pass the end of the clip.
      * per frame, over the periods that started in it, with the last period (and the last difference) of the previous
        frame carried in while the run lasts:
          jitterLocal  = mean |T_i - T_{i-1}| / mean T_i,
          jitterDDP    = mean |(T_i - T_{i-1}) - (T_{i-1} - T_{i-2})| / mean T_i,
          shimmerLocal = mean |A_i - A_{i-1}| / mean A_i,
          logHNR       = ln(c / (1 - c)), c = mean cc* clipped to [1e-3, 1 - 1e-6];
        a voiced frame in which no period starts repeats the values of the previous frame of the run (a first such frame
        gives zeros); unvoiced frames give 0 for all four and end the run (onlyVoiced = 0: the frames are still output)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    T = len(F0)
    out = np.zeros((4, T))
    pos = None
    prevT = prevD = prevA = None
    last = (0.0, 0.0, 0.0, 0.0)
    dead = False
    for t in range(T):
        f = F0[t]
        if not f > 0.0:
            pos = prevT = prevD = prevA = None
            last = (0.0, 0.0, 0.0, 0.0)
            dead = False
            continue
        T0 = P.fs / f
        lo, hi = int(math.ceil((1.0 - JIT_SEARCH_REL) * T0)), int(math.floor((1.0 + JIT_SEARCH_REL) * T0))
        W = _round_half_up(T0)
        start, end = t * P.hop, (t + 1) * P.hop
        if pos is None:
            pos = float(start)
        Ts, As, Cs, dT, dD, dA = [], [], [], [], [], []
        while not dead and pos < end:
            p = int(math.floor(pos))
            if p + W + hi > n or lo < 1 or hi < lo:
                dead = True
                break
            seg = x[p:p + W]
            e0 = float(np.dot(seg, seg))
            lags = np.arange(lo, hi + 1)
            cc = np.zeros(len(lags))
            for q, tau in enumerate(lags):
                s2 = x[p + tau:p + tau + W]
                e1 = float(np.dot(s2, s2))
                cc[q] = float(np.dot(seg, s2)) / math.sqrt(e0 * e1) if e0 > 0.0 and e1 > 0.0 else 0.0
            q = int(np.argmax(cc))
            tau = int(lags[q])
            Tp, cs = float(tau), cc[q]
            if 0 < q < len(lags) - 1:
                y1, y2, y3 = cc[q - 1], cc[q], cc[q + 1]
                den = y1 - 2.0 * y2 + y3
                if den < 0.0:
                    dx = 0.5 * (y1 - y3) / den
                    Tp = tau + dx
                    cs = y2 - 0.125 * (y1 - y3) ** 2 / den
            per = x[p:p + tau]
            A = float(per.max() - per.min())
            if prevT is not None:
                d = Tp - prevT
                dT.append(abs(d))
                if prevD is not None:
                    dD.append(abs(d - prevD))
                prevD = d
                dA.append(abs(A - prevA))
            prevT, prevA = Tp, A
            Ts.append(Tp)
            As.append(A)
            Cs.append(cs)
            pos += Tp
        if Ts:
            mT, mA = float(np.mean(Ts)), float(np.mean(As))
            c = min(max(float(np.mean(Cs)), JIT_CC_MIN), JIT_CC_MAX)
            last = (float(np.mean(dT)) / mT if dT else 0.0,
                    float(np.mean(dD)) / mT if dD else 0.0,
                    float(np.mean(dA)) / mA if dA and mA > 0.0 else 0.0,
                    math.log(c / (1.0 - c)))
        out[:, t] = last
    return out


def pitch_chain(x: np.ndarray, mag: np.ndarray, rms: np.ndarray, P: Params = _P16):
    """Rows (F0final, voicingFinalUnclipped, jitterLocal, jitterDDP, shimmerLocal, logHNR) [6, nF] and the
    per-frame candidates [nF, 6, 3]."""
    nf = mag.shape[0]
    cands = np.zeros((nf, SHS_NCAND, 3))
    for t in range(nf):
        cands[t] = shs_candidates(spec_scale(mag[t], P), P)
    F, V = viterbi_smooth(cands)
    F, V = energy_gate(F, V, rms)
    js = jitter_shimmer(x, F, P)
    return np.vstack([F[None], V[None], js]), cands


def lld(x: np.ndarray, P: Params = _P16) -> np.ndarray:
    """38 low-level descriptors per frame for one clip: float64 [NLLD, nF]."""
    x = np.asarray(x)
    fr, win, mag = magnitudes(x, P)
    nf = fr.shape[0]
    out = np.full((NLLD, nf), np.nan, dtype=np.float64)
    if nf == 0:
        return out
    ham = P.hamming()
    N = P.frame
    # --- cEnergy rms on winframe (Androids.conf:117-123)
    out[0] = np.sqrt(np.sum(win * win, axis=1) / N)
    # --- cMelspec + cMfcc (Androids.conf:101-115)
    melspec = (mag * HTK_SCALE) @ P.mel_matrix().T
    logmel = np.log(np.maximum(melspec, MEL_FLOOR))
    out[1:13] = (logmel @ dct_matrix().T).T
    # --- cMZcr zcr on raw frames (Androids.conf:125-132): sign changes / N
    out[13] = np.sum(fr[:, 1:] * fr[:, :-1] < 0.0, axis=1) / N
    # --- cIntensity on winframe (Androids.conf:134-139)
    im = np.sum(ham[None, :] * win * win, axis=1) / np.sum(ham)
    out[16] = im / I0
    out[17] = (im / I0) ** 0.3
    # --- cSpectral on fftmag (Androids.conf:258-280), power spectrum (squareInput default)
    NB = P.nbins
    Pw = mag * mag
    f = np.arange(NB, dtype=np.float64) * P.df
    tot = np.sum(Pw, axis=1)
    safe = np.where(tot > 0, tot, 1.0)
    out[22] = np.sum(Pw[:, (f >= 250.0) & (f <= 650.0)], axis=1)
    out[23] = np.sum(Pw[:, (f >= 1000.0) & (f <= 4000.0)], axis=1)
    cum = np.cumsum(Pw, axis=1)
    for j, p in enumerate((0.25, 0.50, 0.75, 0.90)):
        # first bin whose inclusive cumulative sum reaches p*total
        idx = np.argmax(cum >= (p * tot)[:, None], axis=1)
        out[24 + j] = idx * P.df
    d = np.diff(mag, axis=0, prepend=mag[:1])
    out[28] = np.sqrt(np.sum(d * d, axis=1) / NB)          # flux (0 for the first frame)
    cen = np.sum(Pw * f[None, :], axis=1) / safe
    out[29] = cen
    p = Pw / safe[:, None]
    out[30] = -np.sum(np.where(p > 0, p * np.log2(np.where(p > 0, p, 1.0)), 0.0), axis=1)
    dev = f[None, :] - cen[:, None]
    var = np.sum(dev ** 2 * p, axis=1)
    out[31] = var
    vs = np.where(var > 0, var, 1.0)
    out[32] = np.sum(dev ** 3 * p, axis=1) / vs ** 1.5
    out[33] = np.sum(dev ** 4 * p, axis=1) / vs ** 2
    sf, sff = np.sum(f), np.sum(f * f)
    out[34] = (NB * np.sum(Pw * f[None, :], axis=1) - sf * tot) / (NB * sff - sf * sf)
    out[35] = np.sum(Pw * P.sharpness_weights()[None, :], axis=1) / safe
    mid = mag[:, 1:-1]
    # excess of every bin over the mean of its two neighbours, clipped at zero: at a spectral peak this is the
    # peak's prominence; unlike a "bin is a strict local maximum" test it is continuous in the magnitudes, so
    # float32 and float64 runs cannot disagree on a decision
    pk = np.sum(np.maximum(mid - 0.5 * (mag[:, :-2] + mag[:, 2:]), 0.0), axis=1)
    msum = np.sum(mag, axis=1)
    out[36] = pk / np.where(msum > 0, msum, 1.0)                 # harmonicity proxy (free choice)
    # geometric / arithmetic mean of the power spectrum; an all-zero frame takes the limit value 1 EXACTLY (free choice:
    # the formula itself gives 1 +- 1e-14 there, and which of several silent frames holds the contour's maximum would
    # then be rounding noise - maxPos must not depend on that)
    out[37] = np.where(tot > 0, np.exp(np.mean(np.log(np.maximum(Pw, 1e-30)), axis=1)) / np.maximum(tot / NB, 1e-30), 1.0)
    # --- cSpecScale .. cPitchJitter (Androids.conf:142-255)
    rows, _ = pitch_chain(np.asarray(x, dtype=np.float64), mag, out[0], P)
    out[I_F0], out[I_VOICE] = rows[0], rows[1]
    out[I_JL:I_HNR + 1] = rows[2:]
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


def functionals(lld_c: np.ndarray, window_frames: int = 0) -> np.ndarray:
    """LLD [38, nF] -> 912 functionals in cCsvSink order.  ``window_frames`` = 0: over the whole clip (adopted reading
    of Androids.conf:349-356, see the module docstring); > 0: over the first ``window_frames`` frames of the
    full-length sma / delta contours (the literal reading, row 0 of the multi-row output)."""
    if lld_c.shape[1] == 0:
        return np.full(NLLD * 2 * NFUNC, np.nan)
    s = sma3(lld_c)
    d = delta2(s)
    if window_frames > 0:
        s, d = s[:, :window_frames], d[:, :window_frames]
    with np.errstate(invalid="ignore"):
        fs = functionals12(s)
        fd = functionals12(d)
    parts = []
    for lo, hi in LEVELS:
        parts.append(fs[lo:hi].reshape(-1))
        parts.append(fd[lo:hi].reshape(-1))
    return np.concatenate(parts)


def extract(x: np.ndarray, fs: int = 16000, window_frames: int = 0) -> np.ndarray:
    """One clip (float32 samples in [-1,1)) -> 912 features."""
    return functionals(lld(x, Params(fs)), window_frames)
