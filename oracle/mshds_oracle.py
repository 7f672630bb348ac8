"""CPU restatement of the Praat analyses behind ``src/mshds_extractor.py``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: the arithmetic lives in
praat-parselmouth 0.4.6 (Praat's C++), which is absent from /root/reference and from this image,
and the reference records no outputs for inputs we hold.  What IS pinned by the reference are the
call order, the parameters and the Python-side post-processing (file:line cited below).  The
algorithms are restated from their publications: Boersma (1993) "Accurate short-term analysis of
the fundamental frequency and the harmonics-to-noise ratio of a sampled sound" (autocorrelation /
cross-correlation pitch, path finder, HNR) and the Praat manual pages "Sound: To Intensity...",
"Sound: To Pitch (ac)...", "Sound: To Harmonicity (cc)...", "Sound: To Spectrogram...",
"Spectrum: Get centre of gravity / central moment...".  Free choices are documented inline.

Built so far (the rest of the 25 features is NaN, as in ``csrc/mshds.hip``):
  a2 ``_speechrate``, a9 ``_measureFormants``, a3 ``_pitch_values``, a4 ``_extract_pitch``, a5 ``_extract_intensity``, a6 ``_extract_harmonicity``,
  a10 ``_extract_Spectral_Moments``, a7 ``_extract_Slope_Tilt``, a8 ``_extract_CPP``.
Arithmetic: float64 on the float32 samples (Praat computes in double).
"""
from __future__ import annotations

import numpy as np

FS = 16000.0
DX = 1.0 / FS

FEATURE_NAMES = [
    "Speaking_Rate", "Articulation_Rate", "Phonation_Ratio", "Pause_Rate", "Mean_Pause_Duration",
    "mean_F0", "stdev_F0_Semitone", "mean_dB", "range_ratio_dB", "HNR_dB",
    "Spectral_Slope", "Spectral_Tilt", "Cepstral_Peak_Prominence",
    "mean_F1_Loc", "std_F1_Loc", "mean_B1_Loc", "std_B1_Loc",
    "mean_F2_Loc", "std_F2_Loc", "mean_B2_Loc", "std_B2_Loc",
    "Spectral_Gravity", "Spectral_Std_Dev", "Spectral_Skewness", "Spectral_Kurtosis",
]                                                          # src/mshds_extractor.py:397-404
BUILT = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24]


# ---- Sampled helpers ------------------------------------------------------------------------------
# A Praat Sound carries its own time axis: x1 = time of the first sample, [xmin = 0, xmax] = its domain.  Read from a
# file: x1 = 0.5 dx, xmax = n / fs.  After Sound_resample (src/mshds_extractor.py:418-419) the domain is still the
# ORIGINAL file's [0, n_in / fs_in] and the 16 kHz grid is centred in it: nx = round(xmax * 16000),
# x1 = (xmax - (nx - 1) dx) / 2, i.e. up to a quarter sample away from 0.5 dx, and xmax != nx dx.  Every analysis below
# takes ``x1`` (and ``xmax`` where Praat uses the domain rather than the physical duration nx dx).
X1_FILE = 0.5 * DX


def short_term_frames(n_samples, window_duration, time_step, x1=X1_FILE):
    """Sampled_shortTermAnalysis: (number of frames, time of the first frame); frames are centred in the PHYSICAL
    extent of the samples (nx dx around x1 - dx/2 + nx dx / 2), whatever the domain."""
    duration = n_samples * DX
    if window_duration > duration:
        return 0, 0.0
    nf = int(np.floor((duration - window_duration) / time_step)) + 1
    mid = x1 - 0.5 * DX + 0.5 * duration
    t1 = mid - 0.5 * nf * time_step + 0.5 * time_step
    return nf, t1


def x_to_low_index(t, x1=X1_FILE):
    """Sampled_xToLowIndex for the sound, 0-based.  Praat works on the 1-based real index (x - x1) / dx + 1 and frame
    times sit on half-sample positions, where the last bit decides: the + 1.0 is kept as its own rounded operation."""
    return np.floor((np.asarray(t) - x1) / DX + 1.0).astype(np.int64) - 1


def x_to_nearest_index(t, x1=X1_FILE):
    """Sampled_xToNearestIndex, 0-based: Melder_iround (= floor (. + 0.5)) of the 1-based real index."""
    return np.floor(((np.asarray(t) - x1) / DX + 1.0) + 0.5).astype(np.int64) - 1


def x_to_high_index(t, x1=X1_FILE):
    return np.ceil((np.asarray(t) - x1) / DX + 1.0).astype(np.int64) - 1


def file_xmax(n_samples):
    """Domain end of a sound of n samples read from a 16 kHz file (Praat: numberOfSamples / sampleRate)."""
    return n_samples / FS


# ---- Intensity (Praat manual "Sound: To Intensity...") -------------------------------------------
def _bessel_i0(x):
    return np.i0(x)


def intensity(x, minimum_pitch, time_step, subtract_mean=True, x1=X1_FILE):
    """dB contour + first frame time.  Effective window 3.2/minimum_pitch (physical 6.4/minimum_pitch),
    Kaiser-20 window (sidelobes below -190 dB), per-frame mean subtracted, reference 4e-10 Pa^2."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    phys = 6.4 / minimum_pitch
    if time_step <= 0:
        time_step = 0.8 / minimum_pitch
    half_dur = 0.5 * phys
    half = int(np.floor(half_dur / DX))
    i = np.arange(-half, half + 1)
    xx = i * DX / half_dur
    win = _bessel_i0((2.0 * np.pi * np.pi + 0.5) * np.sqrt(np.maximum(0.0, 1.0 - xx * xx)))
    nf, t1 = short_term_frames(n, phys, time_step, x1)
    out = np.empty(nf)
    for f in range(nf):
        t = t1 + f * time_step
        mid = int(x_to_nearest_index(t, x1))
        lo, hi = max(0, mid - half), min(n - 1, mid + half)
        seg = x[lo:hi + 1]
        w = win[lo - mid + half: hi - mid + half + 1]
        if subtract_mean:
            seg = seg - seg.mean()
        val = np.sum(seg * seg * w) / np.sum(w) / 4.0e-10
        out[f] = -300.0 if val < 1e-30 else 10.0 * np.log10(val)
    return out, t1, time_step


def vector_extremum_parabolic(y, maximum=True):
    """Vector_getMaximum/Minimum with parabolic interpolation (end points count un-interpolated)."""
    y = np.asarray(y, dtype=np.float64)
    s = 1.0 if maximum else -1.0
    z = s * y
    if len(z) == 0:
        return np.nan
    best = max(z[0], z[-1])
    if len(z) > 2:
        m = z[1:-1]
        loc = (m > z[:-2]) & (m >= z[2:])
        dy = 0.5 * (z[2:] - z[:-2])
        d2 = 2.0 * m - z[:-2] - z[2:]
        with np.errstate(divide="ignore", invalid="ignore"):
            imp = np.where(loc & (d2 != 0), m + 0.5 * dy * dy / d2, -np.inf)
        imp = np.where(loc & (d2 == 0), m, imp)
        if loc.any():
            best = max(best, imp.max())
    return s * best


# ---- sinc interpolation (Praat NUM_interpolate_sinc), vectorised over query points ----------------
def interpolate_sinc(y, x, depth):
    """y: [..., n] (0-based samples 0..n-1), x: [...] real positions (0-based).  Raised-cosine
    windowed sinc of `depth` samples to each side, clipped at the array ends."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    n = y.shape[-1]
    x1 = x + 1.0                                         # Praat is 1-based
    midleft = np.floor(x1).astype(np.int64)
    midright = midleft + 1
    d = np.minimum(np.minimum(depth, midright - 1), n - midleft)
    d = np.maximum(d, 0)
    left = midright - d
    right = midleft + d
    res = np.zeros_like(x1)
    k = np.arange(depth)
    # left half: ix = midleft - k  (k < d)
    a0 = np.pi * (x1 - midleft)
    aa0 = a0 / (x1 - left + 1.0)
    daa = np.pi / (x1 - left + 1.0)
    a = a0[..., None] + np.pi * k
    aa = aa0[..., None] + daa[..., None] * k
    sgn = np.where(k % 2 == 0, 1.0, -1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        wgt = 0.5 * np.sin(a0)[..., None] * sgn / a * (1.0 + np.cos(aa))
    idx = np.clip(midleft[..., None] - k - 1, 0, n - 1)
    vals = np.take_along_axis(y, idx, axis=-1) if y.ndim == idx.ndim else y[idx]
    res = res + np.sum(np.where(k < d[..., None], vals * wgt, 0.0), axis=-1)
    a0 = np.pi * (midright - x1)
    aa0 = a0 / (right - x1 + 1.0)
    daa = np.pi / (right - x1 + 1.0)
    a = a0[..., None] + np.pi * k
    aa = aa0[..., None] + daa[..., None] * k
    with np.errstate(divide="ignore", invalid="ignore"):
        wgt = 0.5 * np.sin(a0)[..., None] * sgn / a * (1.0 + np.cos(aa))
    idx = np.clip(midright[..., None] + k - 1, 0, n - 1)
    vals = np.take_along_axis(y, idx, axis=-1) if y.ndim == idx.ndim else y[idx]
    res = res + np.sum(np.where(k < d[..., None], vals * wgt, 0.0), axis=-1)
    # exact sample / out of range
    ex = np.clip(np.round(x).astype(np.int64), 0, n - 1)
    vex = np.take_along_axis(y, ex[..., None], axis=-1)[..., 0] if y.ndim > 1 else y[ex]
    res = np.where((x1 == midleft) | (x1 > n) | (x1 < 1), vex, res)
    return res


GOLD = 0.5 * (3.0 - np.sqrt(5.0))
SQRT_EPS = np.sqrt(np.finfo(np.float64).eps)
BRENT_ITMAX = 60


def improve_maximum_sinc(y, ix, depth=70, tol=1e-10):
    """NUMimproveMaximum with sinc interpolation: Brent's minimiser (the netlib ``fminbr`` form Praat
    uses: golden-section steps with safeguarded parabolic interpolation, tolerance
    sqrt(eps)*|x| + tol/3 on the 1-based position, at most 60 iterations) applied to -sinc on
    [ix-1, ix+1].  Vectorised over candidates with masks; y: [m, n], ix: [m] 0-based positions.
    Returns (position, value) like Praat (the value is the one found during the search)."""
    ix = np.asarray(ix, dtype=np.float64)
    f = lambda x1: -interpolate_sinc(y, x1 - 1.0, depth)        # x1 is 1-based like Praat's index
    a, b = ix + 1.0 - 1.0, ix + 1.0 + 1.0
    v = a + GOLD * (b - a)
    fv = f(v)
    x, w, fx, fw = v.copy(), v.copy(), fv.copy(), fv.copy()
    active = np.ones(ix.shape, dtype=bool)
    for _ in range(BRENT_ITMAX):
        rng = b - a
        mid = 0.5 * (a + b)
        tol_act = SQRT_EPS * np.abs(x) + tol / 3.0
        active &= ~(np.abs(x - mid) + 0.5 * rng <= 2.0 * tol_act)
        if not active.any():
            break
        step = GOLD * np.where(x < mid, b - x, a - x)
        t = (x - w) * (fx - fv)
        q = (x - v) * (fx - fw)
        p = (x - v) * q - (x - w) * t
        q = 2.0 * (q - t)
        p = np.where(q > 0.0, -p, p)
        q = np.abs(q)
        use = (np.abs(x - w) >= tol_act) & (np.abs(p) < np.abs(step * q)) & \
              (p > q * (a - x + 2.0 * tol_act)) & (p < q * (b - x - 2.0 * tol_act))
        with np.errstate(divide="ignore", invalid="ignore"):
            step = np.where(use, p / q, step)
        step = np.where(np.abs(step) < tol_act, np.where(step > 0.0, tol_act, -tol_act), step)
        tt = x + step
        ft = f(tt)
        better = ft <= fx
        # better: shrink towards tt, shift the history
        a_n = np.where(better, np.where(tt < x, a, x), np.where(tt < x, tt, a))
        b_n = np.where(better, np.where(tt < x, x, b), np.where(tt < x, b, tt))
        c1 = ~better & ((ft <= fw) | (w == x))
        c2 = ~better & ~c1 & ((ft <= fv) | (v == x) | (v == w))
        v_n = np.where(better, w, np.where(c1, w, np.where(c2, tt, v)))
        fv_n = np.where(better, fw, np.where(c1, fw, np.where(c2, ft, fv)))
        w_n = np.where(better, x, np.where(c1, tt, w))
        fw_n = np.where(better, fx, np.where(c1, ft, fw))
        x_n = np.where(better, tt, x)
        fx_n = np.where(better, ft, fx)
        a, b = np.where(active, a_n, a), np.where(active, b_n, b)
        v, fv, w, fw = np.where(active, v_n, v), np.where(active, fv_n, fv), np.where(active, w_n, w), np.where(active, fw_n, fw)
        x, fx = np.where(active, x_n, x), np.where(active, fx_n, fx)
    return x - 1.0, -fx


# ---- Pitch (Boersma 1993) --------------------------------------------------------------------------------
class PitchResult:
    def __init__(self, t1, dt, ceiling, freq, strength, ncand, intensity, selected):
        self.t1, self.dt, self.ceiling = t1, dt, ceiling
        self.freq, self.strength, self.ncand = freq, strength, ncand     # [nF, maxc] candidate lists
        self.intensity = intensity
        self.selected = selected                                          # chosen candidate index per frame

    @property
    def n_frames(self):
        return self.freq.shape[0]

    def frequency(self):
        """selected_array['frequency'] (0 = unvoiced)."""
        f = self.freq[np.arange(self.n_frames), self.selected] if self.n_frames else np.zeros(0)
        return f

    def voiced_values(self):
        f = self.frequency()
        return f[(f > 0.0) & (f < self.ceiling)]

    def defined_at(self, t):
        """Pitch 'Get value at time' (Hertz, linear) is defined iff the nearest frame is voiced."""
        t = np.asarray(t, dtype=np.float64)
        n = self.n_frames
        if n == 0:
            return np.zeros(t.shape, bool)
        ireal = (t - self.t1) / self.dt
        ileft = np.floor(ireal)
        phase = ireal - ileft
        near = np.where(phase < 0.5, ileft, ileft + 1).astype(np.int64)
        xmin, xmax = self.t1 - 0.5 * self.dt, self.t1 + (n - 0.5) * self.dt
        ok = (near >= 0) & (near < n)
        f = self.frequency()
        fn = f[np.clip(near, 0, n - 1)]
        return ok & (fn > 0.0) & (fn < self.ceiling)


def _frame_autocorr(frames, nfft):
    spec = np.fft.rfft(frames, n=nfft, axis=1)
    return np.fft.irfft(spec.real ** 2 + spec.imag ** 2, n=nfft, axis=1)


def _sinc_rows(r, fi, pos, depth, block=2048):
    """interpolate_sinc(r[fi[k]], pos[k]) in blocks (keeps the temporaries small)."""
    out = np.empty(len(fi))
    for b in range(0, len(fi), block):
        sl = slice(b, b + block)
        out[sl] = interpolate_sinc(r[fi[sl]], pos[sl], depth)
    return out


def _improve_rows(r, fi, ix, depth, block=2048):
    xm, ym = np.empty(len(fi)), np.empty(len(fi))
    for b in range(0, len(fi), block):
        sl = slice(b, b + block)
        xm[sl], ym[sl] = improve_maximum_sinc(r[fi[sl]], ix[sl], depth)
    return xm, ym


def _candidates(r, offset_lags, dx_lag, min_lag, max_lag, max_cand, voicing_thr, octave_cost, min_pitch,
                brent_ixmax, refine_depth=70):
    """r: [nF, 2*brent_ixmax+1] symmetric normalised correlation, index = lag + brent_ixmax.
    Boersma (1993) steps 3.10-3.11: local maxima above half the voicing threshold, parabolic
    position, sinc-interpolated strength, at most max_cand-1 voiced candidates (the weakest by
    octave-cost-corrected strength is replaced), then refinement of each kept maximum."""
    nF = r.shape[0]
    freq = np.zeros((nF, max_cand))
    stren = np.zeros((nF, max_cand))
    imax = np.zeros((nF, max_cand), dtype=np.int64)
    ncand = np.ones(nF, dtype=np.int64)                  # candidate 0 = unvoiced
    c = brent_ixmax
    hi = min(max_lag - 1, brent_ixmax - 1)
    lags = np.arange(max(min_lag, 2), hi + 1)
    if nF == 0 or lags.size == 0:
        return freq, stren, ncand
    v = r[:, c + lags]
    ok = (v > 0.5 * voicing_thr) & (v > r[:, c + lags - 1]) & (v >= r[:, c + lags + 1])
    fi, li = np.nonzero(ok)                              # frame-major, ascending lag: Praat's loop order
    if fi.size:
        lag = lags[li]
        y0, y1, y2 = r[fi, c + lag - 1], r[fi, c + lag], r[fi, c + lag + 1]
        dr = 0.5 * (y2 - y0)
        d2r = 2.0 * y1 - y0 - y2
        fmax = 1.0 / DX / (lag + dr / d2r)
        st = _sinc_rows(r, fi, c + 1.0 / DX / fmax, 30)
        st = np.where(st > 1.0, 1.0 / st, st)
        counts = np.bincount(fi, minlength=nF)
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        rank = np.arange(fi.size) - first[fi]
        easy = counts[fi] <= max_cand - 1
        freq[fi[easy], rank[easy] + 1] = fmax[easy]
        stren[fi[easy], rank[easy] + 1] = st[easy]
        imax[fi[easy], rank[easy] + 1] = lag[easy]
        ncand = np.where(counts <= max_cand - 1, counts + 1, max_cand)
        for f in np.nonzero(counts > max_cand - 1)[0]:   # frames with more maxima than slots
            n_in = 1
            for k in range(first[f], first[f] + counts[f]):
                if n_in < max_cand:
                    place = n_in
                    n_in += 1
                else:
                    weakest, place = 2.0, 0
                    for z in range(1, max_cand):
                        loc = stren[f, z] - octave_cost * np.log2(min_pitch / freq[f, z])
                        if loc < weakest:
                            weakest, place = loc, z
                    if st[k] - octave_cost * np.log2(min_pitch / fmax[k]) <= weakest:
                        place = 0
                if place:
                    freq[f, place], stren[f, place], imax[f, place] = fmax[k], st[k], lag[k]
    # second pass: refine every voiced candidate by maximising the sinc-interpolated correlation
    fi, ci = np.nonzero(freq > 0.0)
    if fi.size:
        xm, ym = _improve_rows(r, fi, (imax[fi, ci] + c).astype(np.float64), refine_depth)
        lag_real = xm - c
        ym = np.where(ym > 1.0, 1.0 / ym, ym)
        freq[fi, ci] = 1.0 / DX / lag_real
        stren[fi, ci] = ym
    return freq, stren, ncand


def _path_finder(freq, stren, ncand, intens, dt, silence_thr, voicing_thr, octave_cost, octave_jump_cost,
                 vuv_cost, ceiling):
    """Pitch_pathFinder (Viterbi over candidates); returns the selected candidate per frame."""
    nF, maxc = freq.shape
    if nF == 0:
        return np.zeros(0, dtype=np.int64)
    corr = 0.01 / dt
    ojc, vuc = octave_jump_cost * corr, vuv_cost * corr
    valid = np.arange(maxc)[None, :] < ncand[:, None]
    voiceless = ~((freq > 0.0) & (freq < ceiling))
    unv = np.zeros(nF) if silence_thr <= 0 else 2.0 - intens / (silence_thr / (1.0 + voicing_thr))
    unv = voicing_thr + np.maximum(0.0, unv)
    with np.errstate(divide="ignore", invalid="ignore"):
        delta = np.where(voiceless, unv[:, None], stren - octave_cost * np.log2(ceiling / np.where(voiceless, 1.0, freq)))
    delta = np.where(valid, delta, -1e300)
    psi = np.zeros((nF, maxc), dtype=np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        logf = np.where(voiceless, 0.0, np.log2(np.where(voiceless, 1.0, freq)))
    cur = delta[0].copy()
    for f in range(1, nF):
        v1, v2 = voiceless[f - 1][:, None], voiceless[f][None, :]
        tc = np.where(v1 & v2, 0.0, np.where(v1 | v2, vuc, ojc * np.abs(logf[f - 1][:, None] - logf[f][None, :])))
        val = cur[:, None] - tc + delta[f][None, :]
        val = np.where(valid[f - 1][:, None], val, -np.inf)
        psi[f] = np.argmax(val, axis=0)                   # first maximum, like the strict > of Praat
        cur = np.where(valid[f], val[psi[f], np.arange(maxc)], -1e300)
    sel = np.zeros(nF, dtype=np.int64)
    sel[-1] = int(np.argmax(cur))
    for f in range(nF - 1, 0, -1):
        sel[f - 1] = psi[f, sel[f]]
    return sel


def _hanning(n):
    i = np.arange(1, n + 1)
    return 0.5 - 0.5 * np.cos(i * 2.0 * np.pi / (n + 1))


def pitch_ac(x, time_step=0.0, pitch_floor=75.0, max_candidates=15, very_accurate=False, silence_threshold=0.03,
             voicing_threshold=0.45, octave_cost=0.01, octave_jump_cost=0.35, voiced_unvoiced_cost=0.14,
             pitch_ceiling=600.0, x1=X1_FILE):
    """Sound: To Pitch (ac)... (Hanning window of 3 longest periods)."""
    if very_accurate:
        raise NotImplementedError("very_accurate (Gaussian window) is not used by the reference")
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    ppw = 3.0
    dt = time_step if time_step > 0 else ppw / pitch_floor / 4.0
    ceiling = min(pitch_ceiling, 0.5 / DX)
    dt_window = ppw / pitch_floor
    nsamp_period = int(np.floor(1.0 / DX / pitch_floor))
    half_period = nsamp_period // 2 + 1
    nsamp_window = int(np.floor(dt_window / DX))
    half_window = nsamp_window // 2 - 1
    nsamp_window = half_window * 2
    min_lag = max(2, int(np.floor(1.0 / DX / ceiling)))
    max_lag = min(int(np.floor(nsamp_window / ppw)) + 2, nsamp_window)
    nF, t1 = short_term_frames(n, dt_window, dt, x1)
    interp_depth = 0.5
    brent_ixmax = int(np.floor(nsamp_window * interp_depth))
    nfft = 1
    while nfft < nsamp_window * (1 + interp_depth):
        nfft *= 2
    win = _hanning(nsamp_window)
    wr = _frame_autocorr(win[None, :], nfft)[0]
    wr = wr / wr[0]
    xm = x - x.mean()
    global_peak = np.max(np.abs(xm)) if n else 0.0
    maxc = max_candidates
    if nF <= 0:
        e = np.zeros((0, maxc))
        return PitchResult(t1, dt, ceiling, e, e.copy(), np.zeros(0, np.int64), np.zeros(0), np.zeros(0, np.int64))
    t = t1 + np.arange(nF) * dt
    left = x_to_low_index(t, x1)
    right = left + 1
    # local mean over one longest period to each side
    cs = np.concatenate([[0.0], np.cumsum(x)])
    s0 = np.clip(right - nsamp_period, 0, n - 1)
    s1 = np.clip(left + nsamp_period, 0, n - 1)
    local_mean = (cs[s1 + 1] - cs[s0]) / (2 * nsamp_period)
    start = right - half_window
    idx = start[:, None] + np.arange(nsamp_window)[None, :]
    frames = (x[np.clip(idx, 0, n - 1)] - local_mean[:, None]) * win[None, :]
    a = max(0, half_window - half_period)
    b = min(nsamp_window, half_window + half_period)
    local_peak = np.max(np.abs(frames[:, a:b]), axis=1)
    intens = np.where(local_peak > global_peak, 1.0, local_peak / global_peak) if global_peak > 0 else np.zeros(nF)
    ac = _frame_autocorr(frames, nfft)
    r = np.zeros((nF, 2 * brent_ixmax + 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        pos = ac[:, 1:brent_ixmax + 1] / (ac[:, :1] * wr[None, 1:brent_ixmax + 1])
    pos = np.where(ac[:, :1] > 0, pos, 0.0)
    r[:, brent_ixmax] = 1.0
    r[:, brent_ixmax + 1:] = pos
    r[:, :brent_ixmax] = pos[:, ::-1]
    freq, stren, ncand = _candidates(r, 0, DX, min_lag, max_lag, maxc, voicing_threshold, octave_cost, pitch_floor,
                                     brent_ixmax)
    if global_peak == 0:
        ncand[:] = 1
        freq[:] = 0.0
    sel = _path_finder(freq, stren, ncand, intens, dt, silence_threshold, voicing_threshold, octave_cost,
                       octave_jump_cost, voiced_unvoiced_cost, ceiling)
    return PitchResult(t1, dt, ceiling, freq, stren, ncand, intens, sel)


def pitch_cc(x, time_step, pitch_floor, periods_per_window, max_candidates, silence_threshold, voicing_threshold,
             octave_cost, octave_jump_cost, voiced_unvoiced_cost, pitch_ceiling, accurate=False, x1=X1_FILE):
    """Forward cross-correlation pitch (Boersma 1993 §cc; Praat "To Pitch (cc)")."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    ppw = periods_per_window
    dt = time_step if time_step > 0 else ppw / pitch_floor / 4.0
    ceiling = min(pitch_ceiling, 0.5 / DX)
    dt_window = ppw / pitch_floor
    nsamp_period = int(np.floor(1.0 / DX / pitch_floor))
    half_period = nsamp_period // 2 + 1
    nsamp_window = int(np.floor(dt_window / DX))
    half_window = nsamp_window // 2 - 1
    nsamp_window = half_window * 2
    min_lag = max(2, int(np.floor(1.0 / DX / ceiling)))
    max_lag = min(int(np.floor(nsamp_window / ppw)) + 2, nsamp_window)
    nF, t1 = short_term_frames(n, 1.0 / pitch_floor + dt_window, dt, x1)
    brent_ixmax = int(np.floor(nsamp_window * 1.0))
    maxc = max_candidates
    xm = x - x.mean()
    global_peak = np.max(np.abs(xm)) if n else 0.0
    if nF <= 0:
        e = np.zeros((0, maxc))
        return PitchResult(t1, dt, ceiling, e, e.copy(), np.zeros(0, np.int64), np.zeros(0), np.zeros(0, np.int64))
    t = t1 + np.arange(nF) * dt
    left = x_to_low_index(t, x1)
    right = left + 1
    cs = np.concatenate([[0.0], np.cumsum(x)])
    s0 = np.clip(right - nsamp_period, 0, n - 1)
    s1 = np.clip(left + nsamp_period, 0, n - 1)
    local_mean = (cs[s1 + 1] - cs[s0]) / (2 * nsamp_period)
    start_time = t - 0.5 * (1.0 / pitch_floor + dt_window)
    start = np.maximum(x_to_low_index(start_time, x1), 0)
    span = np.minimum(max_lag + nsamp_window, n - start)
    loc_max_lag = span - nsamp_window
    L = max_lag
    r = np.zeros((nF, 2 * brent_ixmax + 1))
    c = brent_ixmax
    r[:, c] = 1.0
    idx = start[:, None] + np.arange(nsamp_window + L + 1)[None, :]
    seg = x[np.clip(idx, 0, n - 1)] - local_mean[:, None]
    seg = np.where(idx < n, seg, 0.0)
    base = seg[:, :nsamp_window]
    sumx2 = np.sum(base * base, axis=1)
    sq = seg * seg
    csq = np.concatenate([np.zeros((nF, 1)), np.cumsum(sq, axis=1)], axis=1)
    for lag in range(1, L + 1):
        prod = np.sum(base * seg[:, lag:lag + nsamp_window], axis=1)
        sumy2 = csq[:, lag + nsamp_window] - csq[:, lag]
        with np.errstate(divide="ignore", invalid="ignore"):
            v = prod / np.sqrt(sumx2 * sumy2)
        v = np.where((lag <= loc_max_lag) & (sumx2 * sumy2 > 0), v, 0.0)
        r[:, c + lag] = v
        r[:, c - lag] = v
    # local peak over half a longest period around the window centre of the FIRST window
    a = max(0, half_window - half_period)
    b = min(nsamp_window, half_window + half_period)
    local_peak = np.max(np.abs(base[:, a:b]), axis=1)
    intens = np.where(local_peak > global_peak, 1.0, local_peak / global_peak) if global_peak > 0 else np.zeros(nF)
    freq, stren, ncand = _candidates(r, 0, DX, min_lag, max_lag, maxc, voicing_threshold, octave_cost, pitch_floor,
                                     brent_ixmax, refine_depth=700 if accurate else 70)
    if global_peak == 0:
        ncand[:] = 1
        freq[:] = 0.0
    sel = _path_finder(freq, stren, ncand, intens, dt, silence_threshold, voicing_threshold, octave_cost,
                       octave_jump_cost, voiced_unvoiced_cost, ceiling)
    return PitchResult(t1, dt, ceiling, freq, stren, ncand, intens, sel)


def harmonicity_cc(x, time_step=0.01, minimum_pitch=75.0, silence_threshold=0.1, periods_per_window=1.0, x1=X1_FILE):
    """Sound: To Harmonicity (cc): dB per frame, -200 for unvoiced frames."""
    p = pitch_cc(x, time_step, minimum_pitch, periods_per_window, 15, silence_threshold, 0.0, 0.0, 0.0, 0.0,
                 0.5 / DX, accurate=True, x1=x1)
    f = p.frequency()
    s = p.strength[np.arange(p.n_frames), p.selected] if p.n_frames else np.zeros(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        db = np.where(s <= 1e-15, -150.0, np.where(s > 1.0 - 1e-15, 150.0, 10.0 * np.log10(s / (1.0 - s))))
    return np.where(f == 0.0, -200.0, db)


# ---- Spectrogram + spectral moments (Praat manual "Sound: To Spectrogram...", "Spectrum: Get ...") ---
def spectrogram_power(x, window_length=0.005, maximum_frequency=5000.0, time_step=0.002, frequency_step=20.0, x1=X1_FILE):
    """Gaussian-window spectrogram: (power density [nF, nBins], t1, time step, frequency step)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    nyq = 0.5 / DX
    phys = 2.0 * window_length
    eff_t = window_length / np.sqrt(np.pi)
    eff_f = 1.0 / eff_t
    tstep = max(time_step, eff_t / 8.0)
    fstep = max(frequency_step, eff_f / 8.0)
    duration = n * DX
    nsamp = int(np.floor(phys / DX))
    half = nsamp // 2 - 1
    nsamp = half * 2
    if phys > duration or half < 1:
        return np.zeros((0, 0)), 0.0, tstep, fstep
    nT = 1 + int(np.floor((duration - phys) / tstep))
    t1 = x1 + 0.5 * ((n - 1) * DX - (nT - 1) * tstep)
    fmax = maximum_frequency if 0 < maximum_frequency <= nyq else nyq
    nfreq = int(np.floor(fmax / fstep))
    nfft = 1
    while nfft < nsamp or nfft < 2 * nfreq * (nyq / fmax):
        nfft *= 2
    bw_samples = max(1, int(np.floor(fstep * DX * nfft)))
    bw_hz = 1.0 / (DX * nfft)
    fstep = bw_samples * bw_hz
    nfreq = int(np.floor(fmax / fstep))
    i = np.arange(1, nsamp + 1)
    phase = (i - 0.5 * (nsamp + 1)) / nsamp
    edge = np.exp(-12.0)
    win = (np.exp(-48.0 * phase * phase) - edge) / (1.0 - edge)
    one_by = 1.0 / np.sum(win * win) / bw_samples
    t = t1 + np.arange(nT) * tstep
    left = x_to_low_index(t, x1)
    start = left + 1 - half
    idx = start[:, None] + np.arange(nsamp)[None, :]
    frames = x[np.clip(idx, 0, n - 1)] * win[None, :]
    spec = np.fft.rfft(frames, n=nfft, axis=1)
    pw = spec.real ** 2 + spec.imag ** 2
    out = np.empty((nT, nfreq))
    for b in range(nfreq):
        out[:, b] = pw[:, b * bw_samples:(b + 1) * bw_samples].sum(axis=1) * one_by
    return out, t1, tstep, fstep


def spectral_moments(power, fstep):
    """CoG, SD, skewness, kurtosis (power = 2) of each spectrum slice; NaN where the slice is empty."""
    f = np.arange(power.shape[1]) * fstep
    tot = power.sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        cog = (power * f[None, :]).sum(axis=1) / tot
        d = f[None, :] - cog[:, None]
        m2 = (power * d ** 2).sum(axis=1) / tot
        m3 = (power * d ** 3).sum(axis=1) / tot
        m4 = (power * d ** 4).sum(axis=1) / tot
        sd = np.sqrt(m2)
        skew = m3 / (m2 * np.sqrt(m2))
        kurt = m4 / (m2 * m2) - 3.0
    return cog, sd, skew, kurt


# ---- the reference's helpers ----------------------------------------------------------------------------
def pitch_values(x, x1=X1_FILE):
    """``_pitch_values`` (src/mshds_extractor.py:127-162) -> (floor, ceiling)."""
    try:
        p = pitch_ac(x, time_step=0.005, pitch_floor=50.0, pitch_ceiling=600.0, x1=x1)       # :143
        v = p.frequency()
        v = v[v != 0]
        if len(v) == 0:
            return 75, 500
        z = (v - np.mean(v)) / np.std(v)
        v = v[np.abs(z) <= 2]
        if len(v) == 0:
            return 75, 500
        return (60, 250) if np.mean(v) < 170 else (100, 500)
    except Exception:
        return 75, 500


def extract_pitch(x, floor, ceiling, frame_shift=0.005, pitch=None, x1=X1_FILE):
    """``_extract_pitch`` (:164-183): mean F0 (Hz), SD in semitones re 100 Hz (sample SD)."""
    p = pitch if pitch is not None else pitch_ac(x, time_step=frame_shift, pitch_floor=floor, pitch_ceiling=ceiling, x1=x1)
    v = p.voiced_values()
    mean = np.mean(v) if len(v) else np.nan
    st = 12.0 * np.log2(v / 100.0)
    sd = np.std(st, ddof=1) if len(v) > 1 else np.nan
    return mean, sd


def extract_intensity(x, floor, frame_shift=0.005, x1=X1_FILE):
    """``_extract_intensity`` (:185-205): energy-mean dB and max/min ratio of dB values."""
    db, _, _ = intensity(x, floor, frame_shift, True, x1)
    if len(db) == 0:
        return np.nan, np.nan
    mean_db = 10.0 * np.log10(np.mean(10.0 ** (db / 10.0)))
    mn = vector_extremum_parabolic(db, maximum=False)
    mx = vector_extremum_parabolic(db, maximum=True)
    return mean_db, (mx / mn if mn != 0 else np.nan)


def extract_harmonicity(x, floor, ceiling, frame_shift=0.005, x1=X1_FILE):
    """``_extract_harmonicity`` (:207-225): mean HNR over voiced frames."""
    h = harmonicity_cc(x, frame_shift, floor, 0.1, 4.5, x1)
    v = h[h != -200.0]
    return np.mean(v) if len(v) else np.nan


def extract_spectral_moments(x, floor, ceiling, window_size=0.025, frame_shift=0.005, pitch=None, x1=X1_FILE):
    """``_extract_Spectral_Moments`` (:340-376): mean of the 4 moments over frames whose time has a
    defined pitch value."""
    p = pitch if pitch is not None else pitch_ac(x, time_step=frame_shift, pitch_floor=floor, pitch_ceiling=ceiling, x1=x1)
    pw, t1, tstep, fstep = spectrogram_power(x, window_size, 5000.0, frame_shift, 20.0, x1)
    if pw.shape[0] == 0:
        return (np.nan,) * 4
    t = t1 + np.arange(pw.shape[0]) * tstep
    keep = p.defined_at(t)
    cog, sd, sk, ku = spectral_moments(pw[keep], fstep)
    out = []
    for a in (cog, sd, sk, ku):
        a = a[~np.isnan(a)]
        out.append(np.mean(a) if len(a) else np.nan)
    return tuple(out)


# ---- _speechrate (de Jong & Wempe 2009 syllable nuclei, src/mshds_extractor.py:11-125) ---------------
def quantile_sorted(a, factor):
    """Praat NUMquantile on sorted data."""
    n = len(a)
    if n < 1:
        return 0.0
    if n == 1:
        return float(a[0])
    place = factor * n + 0.5
    left = int(np.floor(place))
    left = max(1, min(left, n - 1))
    if a[left] == a[left - 1]:
        return float(a[left - 1])
    return float(a[left - 1] + (place - left) * (a[left] - a[left - 1]))


def value_cubic(y, ireal):
    """Vector 'Get value at time ... Cubic': NUM_interpolate_sinc with depth 2 (0-based real index)."""
    n = len(y)
    x1 = ireal + 1.0
    if x1 > n:
        return float(y[-1])
    if x1 < 1:
        return float(y[0])
    midleft = int(np.floor(x1))
    if x1 == midleft:
        return float(y[midleft - 1])
    midright = midleft + 1
    depth = min(2, midright - 1, n - midleft)
    if depth <= 0:
        return float(y[int(np.floor(x1 + 0.5)) - 1])
    yl, yr = y[midleft - 1], y[midright - 1]
    if depth == 1:
        return float(yl + (x1 - midleft) * (yr - yl))
    dyl = 0.5 * (yr - y[midleft - 2])
    dyr = 0.5 * (y[midright] - yl)
    fil, fir = x1 - midleft, midright - x1
    return float(yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2.0 * (yr - yl))))


def detect_silences(db, t1, dt, xmin, xmax, silence_threshold_db, min_silence, min_sounding):
    """Intensity: To TextGrid (silences): list of [tmin, tmax, is_sounding].  Frames below
    (parabolic maximum - |threshold|) are silent; boundaries sit half-way between frames; then short
    sounding intervals, and after that short silent intervals, are cut (Praat's order)."""
    n = len(db)
    mx = vector_extremum_parabolic(db, True)
    mn = vector_extremum_parabolic(db, False)
    thr = mx - abs(silence_threshold_db)
    if min_silence > (xmax - xmin) or thr < mn or n == 0:
        return [[xmin, xmax, True]]
    iv = []
    start, state = xmin, bool(db[0] < thr)              # state: in silence
    for i in range(1, n):
        sil = bool(db[i] < thr)
        if sil != state:
            tb = t1 + (i - 0.5) * dt
            iv.append([start, tb, not state])
            start, state = tb, sil
    iv.append([start, xmax, not state])

    def cut_short(label_sounding, mindur):
        i = 0
        while i < len(iv):
            a, b, lab = iv[i]
            if lab == label_sounding and (b - a) < mindur and len(iv) > 1:
                del iv[i]
                if i == 0:
                    iv[0][0] = a                           # next interval extended to the left
                elif i == len(iv):
                    iv[-1][1] = b                          # previous extended to the right
                else:
                    iv[i - 1][1] = b                       # previous extended to the right
            else:
                i += 1

    def merge(label_sounding):
        i = 0
        while i < len(iv) - 1:
            if iv[i][2] == label_sounding and iv[i + 1][2] == iv[i][2]:
                a = iv[i][0]
                del iv[i]
                iv[i][0] = a
            else:
                i += 1

    cut_short(True, min_sounding)
    merge(False)
    cut_short(False, min_silence)
    merge(True)
    return iv


def speechrate(x, x1=X1_FILE, xmax=None):
    """``_speechrate`` (:11-125) -> (Speaking_Rate, Articulation_Rate, Phonation_Ratio, Pause_Rate,
    Mean_Pause_Dur).  The harmonicity call of :36-38 is not evaluated, and that changes nothing: (1) its value
    only feeds a no-op (mindip = 2 either way; an undefined mean compares False); (2) its failure path (:123-124,
    all five NaN) is contained in the failure path of the very next call: ``to_harmonicity_cc()`` raises when no cc frame
    fits, i.e. for duration < 1/75 + 1/75 s = 26.7 ms (Sampled_shortTermAnalysis of a 1-period window + 1 period),
    ``to_intensity(50 Hz)`` (:41) raises when duration < 6.4 / 50 = 128 ms, and both land in the same ``except``
    (``harmonicity_failure_implies_intensity_failure`` checks the containment on the frame grids)."""
    nan5 = (np.nan,) * 5
    x = np.asarray(x, dtype=np.float64)
    silencedb, mindip, minpause = -25.0, 2.0, 0.3
    db, t1, dt = intensity(x, 50.0, 0.016, True, x1)                               # :41
    n = len(db)
    if n == 0:
        return nan5                                                                # Praat raises -> :124
    duration = file_xmax(len(x)) if xmax is None else xmax                         # the Intensity keeps the sound's domain
    min_int = vector_extremum_parabolic(db, False)                                 # :42
    max_int = vector_extremum_parabolic(db, True)                                  # :43
    q99 = quantile_sorted(np.sort(db), 0.99)                                       # :47
    silencedb_1 = q99 + silencedb
    if silencedb_1 < min_int:
        silencedb_1 = min_int
    silencedb_2 = silencedb - (max_int - q99)                                      # :51-52
    iv = detect_silences(db, t1, dt, 0.0, duration, silencedb_2, minpause, 0.1)    # :55
    sounding = [(a, b) for a, b, lab in iv if lab]
    npauses = len(sounding)
    if npauses == 0:
        return nan5
    phonation = sum(b - a for a, b in sounding)
    begin_speak, end_speak = sounding[0][0], sounding[-1][1]
    # intensity maxima with sinc-70 interpolated times (:76-81), cubic values (:85)
    idx = [i for i in range(1, n - 1) if db[i] > db[i - 1] and db[i] >= db[i + 1]]
    timepeaks, ints = [], []
    if idx:
        xm, _ = improve_maximum_sinc(np.repeat(db[None, :], len(idx), axis=0), np.array(idx, dtype=np.float64), 70)
        for ir in xm:
            v = value_cubic(db, ir)
            if v > silencedb_1:                                                    # :86
                ints.append(v)
                timepeaks.append(t1 + ir * dt)
    validtime = []
    if len(timepeaks) > 1:                                                         # :92-101
        currenttime, currentint = timepeaks[0], ints[0]
        for p in range(len(timepeaks) - 1):
            nxt = timepeaks[p + 1]
            imin = max(0, int(np.ceil((currenttime - t1) / dt)))
            imax = min(n - 1, int(np.floor((nxt - t1) / dt)))
            if imin <= imax:
                dip = db[imin:imax + 1].min()
            else:
                dip = min(db[min(n - 1, max(0, int(np.floor((currenttime - t1) / dt + 0.5))))],
                          db[min(n - 1, max(0, int(np.floor((nxt - t1) / dt + 0.5))))])
            if abs(currentint - dip) > mindip:
                validtime.append(timepeaks[p])
            currenttime = nxt
            currentint = value_cubic(db, (nxt - t1) / dt)
    pitch = pitch_ac(x, 0.02, 30.0, 4, False, 0.03, 0.25, 0.01, 0.35, 0.25, 450.0, x1=x1)   # :104
    nsyll = 0
    for tm in validtime:                                                           # :106-111
        lab = None
        for k, (a, b, l) in enumerate(iv):
            if (a <= tm < b) or (k == len(iv) - 1 and tm == b):
                lab = l
                break
        if lab and bool(pitch.defined_at(np.array(tm))):
            nsyll += 1
    original_dur = end_speak - begin_speak
    speaking = nsyll / original_dur if original_dur > 0 else 0
    artic = nsyll / phonation if phonation > 0 else 0
    phon_ratio = phonation / original_dur if original_dur > 0 else 0
    n_pauses = npauses - 1
    pause_time = original_dur - phonation
    pause_rate = n_pauses / original_dur if original_dur > 0 else 0
    mean_pause = pause_time / n_pauses if n_pauses > 0 else 0
    return speaking, artic, phon_ratio, pause_rate, mean_pause


# ---- glottal pulses: Sound & Pitch: To PointProcess (cc) -----------------------------------------------
def harmonicity_failure_implies_intensity_failure(n_samples: int) -> bool:
    """True when ``to_harmonicity_cc()`` (defaults: 75 Hz, 1 period per window) yielding no frame (Praat raises) implies
    that ``to_intensity(50 Hz, 0.016)`` yields none either, for a clip of ``n_samples`` samples."""
    nh, _ = short_term_frames(n_samples, 1.0 / 75.0 + 1.0 / 75.0, 0.01)
    ni, _ = short_term_frames(n_samples, 6.4 / 50.0, 0.016)
    return nh > 0 or ni == 0


def pitch_value_at(p, t):
    """Pitch 'Get value at time' (Hertz, linear); NaN where undefined."""
    n = p.n_frames
    if n == 0:
        return np.nan
    f = p.frequency()
    ireal = (t - p.t1) / p.dt
    ileft = int(np.floor(ireal))
    phase = ireal - ileft
    if phase < 0.5:
        inear, ifar = ileft, ileft + 1
    else:
        inear, ifar, phase = ileft + 1, ileft, 1.0 - phase
    if inear < 0 or inear >= n:
        return np.nan
    fn = f[inear]
    if not (0.0 < fn < p.ceiling):
        return np.nan
    if ifar < 0 or ifar >= n:
        return fn
    ff = f[ifar]
    if not (0.0 < ff < p.ceiling):
        return fn
    return fn + phase * (ff - fn)


def _find_extremum(x, tmin, tmax, x1=X1_FILE):
    """Sound_findExtremum (absolute extremum, parabolic position)."""
    n = len(x)
    imin = max(0, int(x_to_low_index(tmin, x1)))
    imax = min(n - 1, int(x_to_high_index(tmax, x1)))
    cnt = imax - imin + 1
    if cnt <= 0:
        return 0.5 * (tmin + tmax)
    seg = x[imin:imax + 1]
    if cnt == 1:
        ie = 1.0
    elif cnt == 2:
        a, b = abs(seg[0]), abs(seg[1])
        ie = 1.0 if a > b else (2.0 if a < b else 1.5)
    else:
        jmin, jmax = int(np.argmin(seg)), int(np.argmax(seg))
        mn, mx = seg[jmin], seg[jmax]
        if mn == mx:
            ie = 0.5 * (cnt + 1.0)
        else:
            j = jmin if abs(mn) > abs(mx) else jmax
            if j == 0:
                ie = 1.0
            elif j == cnt - 1:
                ie = float(cnt)
            else:
                vm, vl, vr = seg[j], seg[j - 1], seg[j + 1]
                ie = (j + 1) + 0.5 * (vr - vl) / (2.0 * vm - vl - vr)
    return x1 + (imin + ie - 1.0) * DX


def _max_correlation(x, t1, window, tmin2, tmax2, x1=X1_FILE):
    """Sound_findMaximumCorrelation -> (correlation, tout, peak)."""
    n = len(x)
    half = 0.5 * window
    ileft1 = int(x_to_nearest_index(t1 - half, x1))
    iright1 = int(x_to_nearest_index(t1 + half, x1))
    ileft2min = int(x_to_low_index(tmin2 - half, x1))
    ileft2max = int(x_to_high_index(tmax2 - half, x1))
    best, r1, r2, r3 = -1.0, 0.0, 0.0, 0.0
    r1b = r3b = 0.0
    ir = 0.0
    peak = 0.0
    i1 = np.arange(ileft1, iright1 + 1)
    for ileft2 in range(ileft2min, ileft2max + 1):
        i2 = i1 - ileft1 + ileft2
        ok = (i1 >= 0) & (i1 < n) & (i2 >= 0) & (i2 < n)
        a1, a2 = x[i1[ok]], x[i2[ok]]
        norm1, norm2, prod = np.dot(a1, a1), np.dot(a2, a2), np.dot(a1, a2)
        local_peak = np.max(np.abs(a2)) if a2.size else 0.0
        r1, r2 = r2, r3
        r3 = prod / np.sqrt(norm1 * norm2) if prod != 0.0 else 0.0
        if r2 > best and r2 >= r1 and r2 >= r3:
            r1b, best, r3b = r1, r2, r3
            ir = ileft2 - 1
            peak = local_peak
    tout = t1
    if best > -1.0:
        d2r = 2.0 * best - r1b - r3b
        if d2r != 0.0:
            dr = 0.5 * (r3b - r1b)
            best += 0.5 * dr * dr / d2r
            ir += dr / d2r
        tout = t1 + (ir - ileft1) * DX
    return best, tout, peak


def point_process_cc(x, p, x1=X1_FILE, xmax=None):
    """Pulse times (Praat Sound_Pitch_to_PointProcess_cc).  The voiced stretches are clipped to the Pitch's domain, which
    is the sound's [0, xmax]."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    duration = file_xmax(n) if xmax is None else xmax
    f = p.frequency()
    voiced = (f > 0.0) & (f < p.ceiling)
    nF = p.n_frames
    pts = []
    global_peak = np.max(np.abs(x)) if n else 0.0
    t = 0.0
    added_right = -1e308
    while True:
        il = int(np.ceil((t - p.t1) / p.dt))                      # Sampled_xToHighIndex, 0-based
        il = max(il, 0)
        while il < nF and not voiced[il]:
            il += 1
        if il >= nF:
            break
        irr = il
        while irr < nF and voiced[irr]:
            irr += 1
        irr -= 1
        tleft = p.t1 + il * p.dt - 0.5 * p.dt
        tright = p.t1 + irr * p.dt + 0.5 * p.dt
        if tleft >= duration - 0.5 * p.dt:
            break
        tleft, tright = max(tleft, 0.0), min(tright, duration)
        tmid = 0.5 * (tleft + tright)
        f0mid = pitch_value_at(p, tmid)
        if np.isnan(f0mid):
            t = tright
            continue
        tmax = _find_extremum(x, tmid - 0.5 / f0mid, tmid + 0.5 / f0mid, x1)
        pts.append(tmax)
        tsave = tmax
        while True:                                              # to the left
            f0 = pitch_value_at(p, tmax)
            if np.isnan(f0):
                break
            corr, tmax, peak = _max_correlation(x, tmax, 1.0 / f0, tmax - 1.25 / f0, tmax - 0.8 / f0, x1)
            if corr == -1.0:
                tmax -= 1.0 / f0
            if tmax < tleft:
                if corr > 0.7 and peak > 0.023333 * global_peak and tmax - added_right > 0.8 / f0:
                    pts.append(tmax)
                break
            if corr > 0.3 and (peak == 0.0 or peak > 0.01 * global_peak):
                if tmax - added_right > 0.8 / f0:
                    pts.append(tmax)
        tmax = tsave
        while True:                                              # to the right
            f0 = pitch_value_at(p, tmax)
            if np.isnan(f0):
                break
            corr, tmax, peak = _max_correlation(x, tmax, 1.0 / f0, tmax + 0.8 / f0, tmax + 1.25 / f0, x1)
            if corr == -1.0:
                tmax += 1.0 / f0
            if tmax > tright:
                if corr > 0.7 and peak > 0.023333 * global_peak:
                    pts.append(tmax)
                    added_right = tmax
                break
            if corr > 0.3 and (peak == 0.0 or peak > 0.01 * global_peak):
                pts.append(tmax)
                added_right = tmax
        t = tright
    return np.sort(np.array(pts, dtype=np.float64))


# ---- resampling to 10 kHz + Formant (burg) -------------------------------------------------------------
RS_DEPTH = 500          # Praat resamples with sinc precision 500 for formant analysis
RS_RATE = 10000.0


def resample_10k(x, x1=X1_FILE, xmax=None):
    """``Sound_resample (me, 10000, 500)`` at the head of Praat's ``Sound_to_Formant_burg``: whole-sound FFT brick-wall
    low-pass, new sample grid centred in the sound's domain, ``NUM_interpolate_sinc`` of depth 500
    (``resample_oracle.sound_resample``)."""
    from .resample_oracle import sound_resample
    x = np.asarray(x, dtype=np.float64)
    return sound_resample(x, x1, DX, 0.0, file_xmax(len(x)) if xmax is None else xmax, RS_RATE, RS_DEPTH)


def _burg(x, m):
    """NUMburg: LPC coefficients a[1..m] (x[n] ~ sum a_k x[n-k])."""
    n = len(x)
    a = np.zeros(m + 1)
    p = np.dot(x, x)
    if p <= 0.0:
        return a[1:]
    b1 = np.zeros(n + 1)
    b2 = np.zeros(n + 1)
    aa = np.zeros(m + 1)
    b1[1] = x[0]
    b2[n - 1] = x[n - 1]
    b1[2:n] = x[1:n - 1]
    b2[1:n - 1] = x[1:n - 1]
    for i in range(1, m + 1):
        num = np.dot(b1[1:n - i + 1], b2[1:n - i + 1])
        den = np.dot(b1[1:n - i + 1], b1[1:n - i + 1]) + np.dot(b2[1:n - i + 1], b2[1:n - i + 1])
        if den <= 0.0:
            return np.zeros(m)
        a[i] = 2.0 * num / den
        for j in range(1, i):
            a[j] = aa[j] - a[i] * aa[i - j]
        if i < m:
            aa[1:i + 1] = a[1:i + 1]
            nb1 = b1[1:n - i] - aa[i] * b2[1:n - i]
            nb2 = b2[2:n - i + 1] - aa[i] * b1[2:n - i + 1]
            b1[1:n - i], b2[1:n - i] = nb1, nb2
    return a[1:]


def formant_burg(x, time_step=0.005, n_formants=5, max_freq=5000.0, half_window=0.025, preemph_from=50.0,
                 safety=50.0, x1=X1_FILE, xmax=None):
    """Sound: To Formant (burg): returns (freq [nF, 5], bw [nF, 5] (NaN padded), t1, dt)."""
    y, x1o, dxo = resample_10k(x, x1, xmax)
    n = len(y)
    nyq = 0.5 / dxo
    npoles = 2 * n_formants
    dt_window = 2.0 * half_window
    duration = n * dxo
    if dt_window > duration:
        return np.zeros((0, n_formants)), np.zeros((0, n_formants)), 0.0, time_step
    nF = int(np.floor((duration - dt_window) / time_step)) + 1
    t1 = x1o - 0.5 * dxo + 0.5 * duration - 0.5 * nF * time_step + 0.5 * time_step
    nsw = int(np.floor(dt_window / dxo))
    half = nsw // 2
    y = y.copy()
    y[1:] = y[1:] - np.exp(-2.0 * np.pi * preemph_from * dxo) * y[:-1]          # Sound_preEmphasis
    i = np.arange(1, nsw + 1)
    imid, edge = 0.5 * (nsw + 1), np.exp(-12.0)
    win = (np.exp(-48.0 * (i - imid) ** 2 / (nsw + 1) ** 2) - edge) / (1.0 - edge)
    F = np.full((nF, n_formants), np.nan)
    B = np.full((nF, n_formants), np.nan)
    for fr in range(nF):
        t = t1 + fr * time_step
        left = int(np.floor((t - x1o) / dxo))
        start, end = left + 1 - half, left + half
        start, end = max(start, 0), min(end, n - 1)
        seg = y[start:end + 1]
        if seg.size < npoles + 2 or np.max(seg * seg) == 0.0:
            continue
        a = _burg(seg * win[:seg.size], npoles)
        poly = np.concatenate([[1.0], -a])                          # z^m - a1 z^(m-1) - ... - am
        r = np.roots(poly)
        for _ in range(3):                                          # polish (Newton) like Praat
            pv = np.polyval(poly, r)
            dv = np.polyval(np.polyder(poly), r)
            r = r - np.where(dv != 0, pv / np.where(dv != 0, dv, 1.0), 0.0)
        big = np.abs(r) > 1.0
        r = np.where(big, 1.0 / np.conj(np.where(big, r, 1.0)), r)    # Roots_fixIntoUnitCircle
        r = r[r.imag >= 0]
        fq = np.abs(np.arctan2(r.imag, r.real)) * nyq / np.pi
        bw = -np.log(r.real ** 2 + r.imag ** 2) * nyq / np.pi
        keep = (fq >= safety) & (fq <= nyq - safety)
        fq, bw = fq[keep], bw[keep]
        o = np.argsort(fq, kind="stable")[:n_formants]
        F[fr, :len(o)] = fq[o]
        B[fr, :len(o)] = bw[o]
    return F, B, t1, time_step


def sampled_value_linear(vals, t1, dt, t):
    """Sampled_getValueAtX with linear interpolation over possibly-undefined (NaN) samples."""
    n = len(vals)
    if n == 0:
        return np.nan
    ireal = (t - t1) / dt
    ileft = int(np.floor(ireal))
    phase = ireal - ileft
    if phase < 0.5:
        inear, ifar = ileft, ileft + 1
    else:
        inear, ifar, phase = ileft + 1, ileft, 1.0 - phase
    if inear < 0 or inear >= n or np.isnan(vals[inear]):
        return np.nan
    if ifar < 0 or ifar >= n or np.isnan(vals[ifar]):
        return vals[inear]
    return vals[inear] + phase * (vals[ifar] - vals[inear])


def measure_formants(x, floor, ceiling, frame_shift=0.005, x1=X1_FILE, xmax=None):
    """``_measureFormants`` (:303-338): F1, B1, F2, B2 at every glottal pulse -> mean and sample SD."""
    x = np.asarray(x, dtype=np.float64)
    F, B, ft1, fdt = formant_burg(x, frame_shift, 5, 5000.0, 0.025, 50.0, x1=x1, xmax=xmax)  # :319
    p = pitch_cc(x, frame_shift, floor, 1.0, 15, 0.03, 0.45, 0.01, 0.35, 0.14, ceiling, x1=x1)   # :320
    pulses = point_process_cc(x, p, x1, xmax)                                                # :321
    lists = [[], [], [], []]
    for t in pulses:                                                                         # :326-331
        for k, arr in enumerate((F[:, 0], B[:, 0], F[:, 1], B[:, 1])):
            v = sampled_value_linear(arr, ft1, fdt, t) if len(arr) else np.nan
            if not np.isnan(v):
                lists[k].append(v)
    out = []
    for l in lists:                                                                          # :333-336
        out.append(np.mean(l) if l else np.nan)
        out.append(np.std(l, ddof=1) if len(l) > 1 else np.nan)
    return tuple(out)


# ---- Ltas (pitch-corrected), slope and tilt  (src/mshds_extractor.py:227-251) ---------------------------
def ltas_pitch_corrected(x, floor, ceiling, max_freq=5000.0, bandwidth=100.0, shortest=0.0001, longest=0.02,
                         max_factor=1.3, x1=X1_FILE, xmax=None):
    """Praat "Sound: To Ltas (pitch-corrected)...": pulses from Sound_to_PointProcess_periodic_cc (AC pitch with
    the standard settings and the automatic time step, then the cc pulse train); every pulse whose two
    neighbouring intervals are plausible periods contributes the energy spectrum of the one period around
    it (rectangular extract, DFT of exactly that many samples); band energies are averaged per band and
    redistributed so that every band weighs the same.  Returns the dB values of the bands or None
    (Praat raises -> the reference returns NaN, NaN)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    p = pitch_ac(x, 0.0, floor, pitch_ceiling=ceiling, x1=x1)
    pulses = point_process_cc(x, p, x1, xmax)
    nb = int(max_freq / bandwidth)
    if len(pulses) - 2 < 1:
        return None
    energy = np.zeros(nb)
    numbers = np.zeros(nb)
    n_periods = 0
    for i in range(1, len(pulses) - 1):
        left, right = pulses[i] - pulses[i - 1], pulses[i + 1] - pulses[i]
        factor = left / right if left > right else right / left
        if not (shortest <= left <= longest and shortest <= right <= longest and factor <= max_factor):
            continue
        t1, t2 = pulses[i] - 0.5 * left, pulses[i] + 0.5 * right
        ix1 = int(np.ceil((t1 - x1) / DX))                       # Sound_extractPart: 1 + ceil / 1 + floor of (t - x1) / dx,
        ix2 = int(np.floor((t2 - x1) / DX))                      # 0-based here; samples outside the sound are zero
        if ix2 < ix1:
            return None                                           # "Extracted Sound would contain no samples"
        m = ix2 - ix1 + 1
        seg = np.zeros(m)
        a, b = max(ix1, 0), min(ix2, n - 1)
        if b >= a:
            seg[a - ix1:b - ix1 + 1] = x[a:b + 1]
        spec = np.fft.rfft(seg) * DX                              # Sound_to_Spectrum (fast = no)
        sdx = 1.0 / (DX * m)
        for k in range(len(spec)):
            band = int(np.ceil(k * sdx / bandwidth))
            if 1 <= band <= nb:
                energy[band - 1] += (spec[k].real ** 2 + spec[k].imag ** 2) * 2.0 * sdx
                numbers[band - 1] += 1
        n_periods += 1
    if n_periods < 1:
        return None
    total = numbers.sum()
    z = np.full(nb, np.nan)
    duration = file_xmax(n) if xmax is None else xmax            # PointProcess_Sound_to_Ltas: sound->xmax - sound->xmin
    for b in range(nb):
        if numbers[b] > 0:
            mean_e = energy[b] / numbers[b]
            z[b] = 10.0 * np.log10(mean_e * (total / nb) / bandwidth / duration / 4.0e-10)
    if np.all(np.isnan(z)):
        return None
    out = z.copy()
    for b in range(nb):                                           # undefined bands: copy / interpolate
        if np.isnan(z[b]):
            bl, br = b - 1, b + 1
            while bl >= 0 and np.isnan(z[bl]):
                bl -= 1
            while br < nb and np.isnan(z[br]):
                br += 1
            if bl < 0:
                out[b] = z[br]
            elif br >= nb:
                out[b] = z[bl]
            else:
                out[b] = ((br - b) * z[bl] + (b - bl) * z[br]) / (br - bl)
    return out


def sampled_mean_rect(z, x1, dx, xmin, xmax):
    """Praat Sampled_getMean without interpolation: every sample is a bar of width dx."""
    nx = len(z)
    xmin, xmax = max(xmin, x1 - 0.5 * dx), min(xmax, x1 + (nx - 0.5) * dx)
    if not xmin < xmax:
        return np.nan
    rimin, rimax = (xmin - x1) / dx + 1.0, (xmax - x1) / dx + 1.0       # 1-based real indices
    total, rng = 0.0, 0.0
    if rimax >= 0.5 and rimin < nx + 0.5:
        imin = 0 if rimin < 0.5 else int(np.floor(rimin + 0.5))
        imax = nx + 1 if rimax >= nx + 0.5 else int(np.floor(rimax + 0.5))
        for i in range(imin + 1, imax):
            if not np.isnan(z[i - 1]):
                rng += 1.0
                total += z[i - 1]
        if imin == imax:
            if 1 <= imin <= nx and not np.isnan(z[imin - 1]):
                ph = rimax - rimin
                rng += ph
                total += ph * z[imin - 1]
        else:
            if imin >= 1 and not np.isnan(z[imin - 1]):
                ph = imin - rimin + 0.5
                rng += ph
                total += ph * z[imin - 1]
            if imax <= nx and not np.isnan(z[imax - 1]):
                ph = rimax - imax + 0.5
                rng += ph
                total += ph * z[imax - 1]
    return total / rng if rng > 0.0 else np.nan


def line_fit_theil_incomplete(xv, yv):
    """Praat NUMlineFit_theil (incomplete method = the "Robust" choice): median of the slopes between
    point i and point i + ceil(n/2)."""
    n = len(xv)
    if n < 2:
        return 0.0
    nc = n // 2
    n2 = nc + 1 if n % 2 == 1 else nc
    slopes = np.sort([(yv[n2 + i] - yv[i]) / (xv[n2 + i] - xv[i]) for i in range(nc)])
    return quantile_sorted(slopes, 0.5)


def extract_slope_tilt(x, floor, ceiling, x1=X1_FILE, xmax=None):
    """src/mshds_extractor.py:227-251: Ltas "Get slope" 50-1000 vs 1000-4000 Hz in dB and the slope of the
    robust line fit over 100-5000 Hz (linear frequency) read from "Report spectral tilt"."""
    z = ltas_pitch_corrected(x, floor, ceiling, x1=x1, xmax=xmax)
    if z is None:
        return np.nan, np.nan
    bw = 100.0
    f1 = 0.5 * bw
    low = sampled_mean_rect(z, f1, bw, 50.0, 1000.0)
    high = sampled_mean_rect(z, f1, bw, 1000.0, 4000.0)
    slope = high - low
    nb = len(z)
    imin = max(1, 1 + int(np.ceil((100.0 - f1) / bw)))
    imax = min(nb, 1 + int(np.floor((5000.0 - f1) / bw)))
    if imax - imin + 1 < 2:
        return np.nan, np.nan                                      # the report raises -> both NaN in the reference
    fx = f1 + (np.arange(imin, imax + 1) - 1) * bw
    tilt = line_fit_theil_incomplete(fx, z[imin - 1:imax])
    return slope, tilt


# ---- Cepstral peak prominence of the voiced stretches  (src/mshds_extractor.py:253-301) ------------------
CPP_FS = 10000.0            # To PowerCepstrogram: 2 * maximum frequency 5000 Hz
CPP_DEPTH = 50              # Sound_resample precision used by Sound_to_PowerCepstrogram
CPP_PITCH_FLOOR = 60.0
CPP_DT = 0.002
CPP_PREEMPH_FROM = 50.0


def vuv_intervals(pulses, xmin, xmax, max_period=0.02, mean_period=0.1):
    """PointProcess: To TextGrid (vuv): the voiced intervals [(tmin, tmax)].  A voiced stretch is a run of
    pulses less than max_period apart, widened by half the mean period on both sides, never before the end
    of the previous stretch nor beyond the sound."""
    out = []
    half = 0.5 * mean_period
    begin_voiceless = xmin
    i, n = 0, len(pulses)
    while i < n:
        end_voiceless = pulses[i] - half
        if end_voiceless <= begin_voiceless:
            end_voiceless = begin_voiceless
        begin_voiced = end_voiceless
        j = i + 1
        while j < n and not (pulses[j] - pulses[j - 1] > max_period):
            j += 1
        j -= 1
        end_voiced = min(pulses[j] + half, xmax)
        out.append((begin_voiced, end_voiced))
        begin_voiceless = end_voiced
        i = j + 1
    return out


def resample_part(seg, x1_seg, duration, fs_out, depth):
    """``Sound_resample`` of an extracted part: ``len(seg)`` samples at 16 kHz, the first one at ``x1_seg``, domain
    ``[0, duration]`` -> (samples, x1, dx): FFT low-pass over the part's own power of two, grid centred in the
    domain, ``NUM_interpolate_sinc`` (``resample_oracle.sound_resample``)."""
    from .resample_oracle import sound_resample
    return sound_resample(np.asarray(seg, dtype=np.float64), x1_seg, DX, 0.0, duration, fs_out, depth)


def power_cepstrogram(seg, x1_seg, duration):
    """Sound: To PowerCepstrogram (60 Hz, 2 ms, 5 kHz, pre-emphasis from 50 Hz) of an extracted part with
    domain [0, duration]: resample to 10 kHz, pre-emphasise, 0.1 s Gaussian-windowed frames (the window shrinks
    to the physical duration of a shorter sound), power spectrum -> ln -> inverse FFT -> squared.
    Returns z [nq, n_frames] (quefrency step 1e-4 s)."""
    n_in = len(seg)
    window = min(2.0 * 3.0 / CPP_PITCH_FLOOR, DX * n_in)
    y, x1o, dxo = resample_part(seg, x1_seg, duration, CPP_FS, CPP_DEPTH)
    a = np.exp(-2.0 * np.pi * CPP_PREEMPH_FROM * dxo)
    y = np.concatenate([y[:1], y[1:] - a * y[:-1]])
    my_duration = DX * n_in                                        # Sampled_shortTermAnalysis on the 16 kHz part
    nf = int(np.floor((my_duration - window) / CPP_DT)) + 1
    mid = x1_seg - 0.5 * DX + 0.5 * my_duration
    t1 = mid - 0.5 * nf * CPP_DT + 0.5 * CPP_DT
    nx = int(np.floor(window * CPP_FS + 0.5))
    nfft = 2
    while nfft < nx:
        nfft *= 2
    nq = nfft // 2 + 1
    i = np.arange(1, nx + 1)
    imid, edge = 0.5 * (nx + 1), np.exp(-12.0)
    win = (np.exp(-48.0 * (i - imid) ** 2 / (nx + 1) ** 2) - edge) / (1.0 - edge)
    z = np.empty((nq, nf))
    sdx = 1.0 / (dxo * nfft)
    m = len(y)
    for f in range(nf):
        t = t1 + f * CPP_DT
        idx0 = int(np.floor((t - 0.5 * window - x1o) / dxo + 0.5))          # Sampled_xToNearestIndex, 0-based
        j = idx0 + np.arange(nx)
        fr = np.where((j >= 0) & (j < m), y[np.clip(j, 0, m - 1)], 0.0)
        fr = (fr - fr.mean()) * win
        spec = np.fft.rfft(fr, nfft) * dxo
        logp = np.log(spec.real ** 2 + spec.imag ** 2 + 1e-300)
        c = np.fft.irfft(logp, nfft) * nfft * sdx                             # unnormalised Hermitian sum times df
        z[:, f] = c[:nq] ** 2
    return z


def _moving_average(v, window):
    """VECsmoothByMovingAverage: [i - w/2, i + w/2] (one less on the right for an even w), clipped."""
    n = len(v)
    out = np.empty(n)
    for i in range(n):
        lo, hi = i - window // 2, i + window // 2
        if window % 2 == 0:
            hi -= 1
        lo, hi = max(lo, 0), min(hi, n - 1)
        out[i] = np.sum(v[lo:hi + 1]) / (hi - lo + 1)
    return out


def cpps(z, dq=1e-4, time_window=0.01, quef_window=0.001, pitch_floor=60.0, pitch_ceiling=330.0):
    """PowerCepstrogram: Get CPPS ("no", 0.01, 0.001, 60, 330, 0.05, parabolic, 0.001, 0, Straight, Robust):
    moving averages over time and quefrency, then per frame the parabolic peak (dB) in [1/330, 1/60] s minus
    the robust (Theil, incomplete) straight trend over the whole quefrency range (an end of 0 resets the
    range), averaged over the frames."""
    nq, nf = z.shape
    zs = z.copy()
    nt = int(np.floor(time_window / CPP_DT))
    if nt > 1:
        for q in range(nq):
            zs[q] = _moving_average(zs[q], nt)
    nqb = int(np.floor(quef_window / dq))
    if nqb > 1:
        for f in range(nf):
            zs[:, f] = _moving_average(zs[:, f], nqb)
    quef = np.arange(nq) * dq
    imin = int(np.ceil((1.0 / pitch_ceiling) / dq))                 # 0-based Sampled_getWindowSamples
    imax = min(int(np.floor((1.0 / pitch_floor) / dq)), nq - 1)
    vals = []
    for f in range(nf):
        db = 10.0 * np.log10(zs[:, f] + 1e-30)
        # trend line: Theil's incomplete method, slope then the median of the residual offsets
        nc = nq // 2
        n2 = nc + 1 if nq % 2 == 1 else nc
        slopes = np.sort((db[n2:n2 + nc] - db[:nc]) / (quef[n2:n2 + nc] - quef[:nc]))
        slope = quantile_sorted(slopes, 0.5)
        icpt = quantile_sorted(np.sort(db - slope * quef), 0.5)
        # Vector_getMaximumAndX with parabolic interpolation
        if imax < imin:
            return np.nan
        peak, xq = db[imin], float(imin)
        if db[imax] > peak:
            peak, xq = db[imax], float(imax)
        lo, hi = max(imin, 1), min(imax, nq - 2)
        for i in range(lo, hi + 1):
            if db[i] > db[i - 1] and db[i] >= db[i + 1]:
                dy = 0.5 * (db[i + 1] - db[i - 1])
                d2y = 2.0 * db[i] - db[i - 1] - db[i + 1]
                loc = db[i] + 0.5 * dy * dy / d2y
                if loc > peak:
                    peak, xq = loc, i + dy / d2y
        qpeak = min(max(xq * dq, 1.0 / pitch_ceiling), 1.0 / pitch_floor)
        vals.append(peak - (slope * qpeak + icpt))
    return float(np.mean(vals)) if vals else np.nan


def extract_cpp(x, floor, ceiling, frame_shift=0.005, x1=X1_FILE, xmax=None):
    """_extract_CPP (:253-301): mean CPPS over the voiced intervals whose CPPS exceeds 4 dB."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    xmax = file_xmax(n) if xmax is None else xmax
    p = pitch_ac(x, frame_shift, floor, voicing_threshold=0.3, pitch_ceiling=ceiling, x1=x1)   # :270
    pulses = point_process_cc(x, p, x1, xmax)                                                # :271
    vals = []
    for (tmin, tmax) in vuv_intervals(pulses, 0.0, xmax):                                    # :272
        tmin, tmax = float(f"{tmin:.6f}"), float(f"{tmax:.6f}")                              # Down to Table, 6 decimals
        if tmin >= tmax:
            continue
        ix1 = int(np.ceil((tmin - x1) / DX))
        ix2 = int(np.floor((tmax - x1) / DX))
        if ix2 < ix1:
            return np.nan                                        # extract_part raises outside the inner try -> NaN (:299-300)
        seg = np.zeros(ix2 - ix1 + 1)
        a, b = max(ix1, 0), min(ix2, n - 1)
        if b >= a:
            seg[a - ix1:b - ix1 + 1] = x[a:b + 1]
        x1_seg = x1 + ix1 * DX - tmin
        z = power_cepstrogram(seg, x1_seg, tmax - tmin)
        v = cpps(z)
        if not np.isnan(v) and v > 4:
            vals.append(v)
    return float(np.mean(vals)) if vals else np.nan


def extract(x, x1=X1_FILE, xmax=None):
    """One clip -> 25 features in the reference's column order.  ``x1`` / ``xmax``: the sound's time axis (first sample,
    end of the domain); the defaults are those of a sound read from a 16 kHz file, ``resample_oracle.resample_praat_sound``
    gives the ones of a file the reference resamples first (:418-419)."""
    x = np.asarray(x, dtype=np.float64)
    xmax = file_xmax(len(x)) if xmax is None else xmax
    out = np.full(25, np.nan)
    out[0:5] = speechrate(x, x1, xmax)                                               # :426
    floor, ceiling = pitch_values(x, x1)                                             # :428
    p = pitch_ac(x, time_step=0.005, pitch_floor=floor, pitch_ceiling=ceiling, x1=x1)   # :178 == :355
    out[5], out[6] = extract_pitch(x, floor, ceiling, 0.005, p)                      # :430
    out[7], out[8] = extract_intensity(x, floor, 0.005, x1)                          # :431
    out[9] = extract_harmonicity(x, floor, ceiling, 0.005, x1)                       # :432
    out[10], out[11] = extract_slope_tilt(x, floor, ceiling, x1, xmax)               # :433
    out[12] = extract_cpp(x, floor, ceiling, 0.005, x1, xmax)                        # :434
    out[13:21] = measure_formants(x, floor, ceiling, 0.005, x1, xmax)                # :441
    out[21:25] = extract_spectral_moments(x, floor, ceiling, 0.025, 0.005, p, x1)    # :446
    return out, (floor, ceiling)
