"""MSHDS (Praat-style) feature extractor on the HIP path (drop-in for ``src/mshds_extractor.py``).

The reference runs ~10 Praat analyses per file through parselmouth, one file at a time, with Python
loops per pulse / per frame (``src/mshds_extractor.py:11-376``).  Here the analyses of a whole batch
run as float64 HIP kernels (``csrc/mshds.hip``); the host only builds the frame grids (Praat's
``Sampled_shortTermAnalysis`` arithmetic), the window tables, and routes each clip to the
speaker-adapted pitch range that ``_pitch_values`` chooses (``:127-162``).

All 25 columns are computed on the device (``csrc/mshds.hip``, ``csrc/mshds_cpp.hip``); there is no CPU
fallback.  A helper that fails in the reference gives NaN for its columns there (``except: return nan``);
the kernels reproduce those cases (too-short clips, no voiced frames, no periods) as NaN as well.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib
from .wavio import read_wav_mono_device

FS = 16000.0
DX = 1.0 / FS
SAMPLE_RATE = 16000

FEATURE_NAMES = [
    "Speaking_Rate", "Articulation_Rate", "Phonation_Ratio", "Pause_Rate", "Mean_Pause_Duration",
    "mean_F0", "stdev_F0_Semitone", "mean_dB", "range_ratio_dB", "HNR_dB",
    "Spectral_Slope", "Spectral_Tilt", "Cepstral_Peak_Prominence",
    "mean_F1_Loc", "std_F1_Loc", "mean_B1_Loc", "std_B1_Loc",
    "mean_F2_Loc", "std_F2_Loc", "mean_B2_Loc", "std_B2_Loc",
    "Spectral_Gravity", "Spectral_Std_Dev", "Spectral_Skewness", "Spectral_Kurtosis",
]                                                            # src/mshds_extractor.py:397-404
BUILT_COLUMNS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24]

CLIP_INFO = np.dtype([("sample_off", "<i8"), ("frame_off", "<i8"), ("t1", "<f8"),
                      ("n_samples", "<i4"), ("n_frames", "<i4"), ("x1", "<f8"), ("xmax", "<f8")])
assert CLIP_INFO.itemsize == 48
RESAMPLE_INFO = np.dtype([("sample_off", "<i8"), ("out_off", "<i8"), ("pos0", "<f8"), ("x1o", "<f8"),
                          ("n_in", "<i4"), ("n_out", "<i4"), ("table", "<i4"), ("pad", "<i4")])
assert RESAMPLE_INFO.itemsize == 48
LP_SIG = np.dtype([("in_off", "<i8"), ("out_off", "<i8"), ("work_off", "<i8"), ("n", "<i4"), ("lg", "<i4")])
assert LP_SIG.itemsize == 32
RS_DEPTH = 500
RS_RATE = 10000.0


def resample10k_tables(pos0: float, depth: int = RS_DEPTH):
    """Weights of ``NUM_interpolate_sinc`` at full depth for the five fractional positions of the 16 kHz -> 10 kHz grid
    (output sample 5q + r sits at the real input index pos0 + 1.6 r + 8q) -> (float64 [5, 2 depth + 1], bases [5]):
    tap j of row r belongs to the input sample bases[r] + 8q + j - depth.  ``depth`` samples to the left of the position
    (taps depth, depth - 1, ...) and ``depth`` to the right (taps depth + 1, ...), each side under its own raised cosine
    that reaches zero one sample beyond its outermost sample."""
    k = np.arange(depth)
    sgn = np.where(k % 2 == 0, 1.0, -1.0)
    rows, bs = [], []
    for r in range(5):
        pr = pos0 + 1.6 * r
        b = int(math.floor(pr))
        f = pr - b
        w = np.zeros(2 * depth + 1)
        if f == 0.0:
            w[depth] = 1.0
        else:
            for a0, span, first, step in ((math.pi * f, f + depth, depth, -1), (math.pi * (1.0 - f), depth + 1.0 - f, depth + 1, 1)):
                a = a0 + math.pi * k
                w[first + step * k] = 0.5 * math.sin(a0) * sgn / a * (1.0 + np.cos(a / span))
        rows.append(w)
        bs.append(b)
    return np.stack(rows), bs


def short_term_frames(n_samples: int, window_duration: float, time_step: float, x1: float = 0.5 * DX):
    """Praat Sampled_shortTermAnalysis of a sound of ``n_samples`` samples whose first sample lies at ``x1``:
    (frames, time of first frame); integer-exact contract.  The physical duration is nx * dx (not the domain)."""
    duration = n_samples * DX
    if window_duration > duration:
        return 0, 0.0
    nf = int(math.floor((duration - window_duration) / time_step)) + 1
    mid = x1 - 0.5 * DX + 0.5 * duration
    t1 = mid - 0.5 * nf * time_step + 0.5 * time_step
    return nf, t1


class SoundDomain:
    """Time domain of every clip of a packed batch, as Praat's Sound carries it: ``x1`` = time of the first sample,
    ``xmax`` = end of the domain [0, xmax].  A sound read from a 16 kHz file has x1 = dx / 2 and xmax = n dx; after
    ``Sound_resample`` the grid is centred in the ORIGINAL domain: xmax = n_in / fs_in, n = round(xmax * 16000),
    x1 = (xmax - (n - 1) dx) / 2 (``src/mshds_extractor.py:418-419``)."""

    def __init__(self, lengths, x1=None, xmax=None):
        self.x1 = [0.5 * DX] * len(lengths) if x1 is None else [float(v) for v in x1]
        self.xmax = [int(n) / FS for n in lengths] if xmax is None else [float(v) for v in xmax]
        assert len(self.x1) == len(lengths) and len(self.xmax) == len(lengths)

    def sub(self, ids):
        return SoundDomain(ids, [self.x1[i] for i in ids], [self.xmax[i] for i in ids])


def _clip_info(sample_offs, lengths, grid, dom=None):
    """grid(n_samples, x1) -> (n_frames, t1).  Returns (structured host array, total frames, max frames)."""
    dom = dom if dom is not None else SoundDomain(lengths)
    ci = np.zeros(len(lengths), dtype=CLIP_INFO)
    off = 0
    for i, (so, n) in enumerate(zip(sample_offs, lengths)):
        nf, t1 = grid(int(n), dom.x1[i])
        ci[i] = (int(so), off, t1, int(n), nf, dom.x1[i], dom.xmax[i])
        off += nf
    return ci, off, int(ci["n_frames"].max()) if len(ci) else 0


def _dev(arr, device, blocking=False):
    """Host array -> device tensor (structured arrays as bytes), queued on the current stream from a pinned staging buffer
    (``blocking``: complete on return - for tables that are cached and later read from other streams).
    A plain ``.to(device)`` of pageable memory is a synchronous copy: the host then waits for everything queued before it,
    about fifty times per batch, and every wait leaves the GPU idle until the next launch arrives."""
    import torch
    a = np.ascontiguousarray(arr)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1)) if a.dtype.fields is not None else torch.from_numpy(a)
    if blocking or torch.device(device).type != "cuda" or t.numel() == 0:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def sinc_cheb_table(depth: int, ncoef: int = 16):
    """Chebyshev coefficients (on frac in [0, 1]) of the weights of Praat's depth-``depth`` sinc interpolation:
    S(b + frac) = sum_o W_o(frac) y[b + o], o = -(depth-1) .. depth.  float64 [2 * depth, ncoef], row o + depth - 1.
    Evaluated at the ``ncoef`` Chebyshev nodes and transformed exactly (DCT); the fit error is < 1e-12."""
    d = int(depth)
    n = np.arange(ncoef)
    nodes = np.cos(np.pi * (n + 0.5) / ncoef)
    frac = ((nodes + 1.0) / 2.0)[:, None]
    hs = 0.5 * np.sin(np.pi * frac)
    kk = np.arange(d)[None, :]
    sign = np.where(kk % 2 == 0, 1.0, -1.0)
    a_l = np.pi * (frac + kk)
    w_l = hs * sign / a_l * (1.0 + np.cos(a_l / (d + frac)))
    a_r = np.pi * (1.0 - frac + kk)
    w_r = hs * sign / a_r * (1.0 + np.cos(a_r / (d + 1.0 - frac)))
    w = np.concatenate([w_l[:, ::-1], w_r], axis=1)                     # [ncoef nodes, 2d taps]
    j = np.arange(ncoef)[:, None]
    t = np.cos(np.pi * j * (n[None, :] + 0.5) / ncoef)
    c = (2.0 / ncoef) * (t @ w)
    c[0] *= 0.5
    return np.ascontiguousarray(c.T)


CHEB_CLIP_MAX_DEPTH = 128


def sinc_cheb_tables(depth: int, with_clipped: bool):
    """The table ``rsaf_mshds_pitch`` takes as ``sinc_cheb``: the full-depth Chebyshev table, and behind it (``with_clipped``)
    the tables of the clipped depths e = 1 .. depth - 1, depth e at offset 2 depth 16 + 16 e (e - 1) doubles."""
    parts = [sinc_cheb_table(depth).reshape(-1)]
    if with_clipped:
        parts += [sinc_cheb_table(e).reshape(-1) for e in range(1, depth)]
    return np.concatenate(parts)


CELL_TABLES_MAX_LDS = 96 * 1024


def sinc_cell_tables(max_lag: int, brent_ixmax: int, min_lag: int, depth: int):
    """Chebyshev tables per CELL for an analysis whose interpolation depth the array ends clip (``rsaf_mshds_pitch`` with
    ``params[17] = 2``; ``csrc/mshds.hip: pitch_cell_coef_kernel``).  A cell is the interval right of sample b of Praat's
    symmetric array r[0 .. 2 ixmax]; its depth is min(depth, b + 1, RN - b - 1) (NUM_interpolate_sinc).  Only the
    samples |lag| <= L are non-zero and r[-lag] = r[lag], so cell b keeps, for every lag m = 0 .. L, the SUM of the two rows
    of its depth's table that meet lag m (taps o = ixmax - b -+ m; one row for m = 0): the coefficients of a cell are
    sum_m table[b][m][j] r[m].  Returns float64 [n_b, ntap_pad, 16] for b = ixmax + lag_lo - 1 .. ixmax + lag_hi
    (lag_lo = max(min_lag, 2), lag_hi = min(max_lag, ixmax) - 1; ntap_pad = L + 1 rounded up to a multiple of 4, zero
    rows behind), or None when a cell's depth falls below 3 (Praat switches to nearest / linear / cubic there) or the
    table does not fit the kernel's LDS."""
    L, RC = int(max_lag), int(brent_ixmax)
    RN = 2 * RC + 1
    lag_lo, lag_hi = max(int(min_lag), 2), min(L - 1, RC - 1)
    if lag_hi < lag_lo:
        return None
    b_lo, b_hi = RC + lag_lo - 1, RC + lag_hi
    ntap_pad = (L + 1 + 3) & ~3
    if ntap_pad * 16 * 8 + 4 * 192 * 4 > CELL_TABLES_MAX_LDS:
        return None
    out = np.zeros((b_hi - b_lo + 1, ntap_pad, 16))
    m = np.arange(L + 1)
    by_depth = {}
    for b in range(b_lo, b_hi + 1):
        dc = min(int(depth), b + 1, RN - b - 1)
        if dc < 3:
            return None
        if dc not in by_depth:
            by_depth[dc] = sinc_cheb_table(dc)
        tab = by_depth[dc]
        for sign in (1, -1):
            o = RC - b + sign * m                           # the tap that meets lag +m / -m
            ok = (o >= -(dc - 1)) & (o <= dc)
            if sign < 0:
                ok &= m > 0
            out[b - b_lo, m[ok]] += tab[o[ok] + dc - 1]
    return out


class _PitchGeom:
    """Window geometry of Sound: To Pitch (ac/cc) for one parameter set (Boersma 1993)."""

    def __init__(self, time_step, floor, ceiling, periods, is_cc):
        self.periods, self.is_cc, self.floor = periods, is_cc, float(floor)
        self.dt = time_step if time_step > 0 else periods / floor / 4.0
        self.ceiling = min(float(ceiling), 0.5 / DX)
        self.dt_window = periods / floor
        self.nsamp_period = int(math.floor(1.0 / DX / floor))
        nsw = int(math.floor(self.dt_window / DX))
        self.half_window = nsw // 2 - 1
        self.nsamp_window = self.half_window * 2
        self.min_lag = max(2, int(math.floor(1.0 / DX / self.ceiling)))
        self.max_lag = min(int(math.floor(self.nsamp_window / periods)) + 2, self.nsamp_window)
        self.brent_ixmax = int(math.floor(self.nsamp_window * (1.0 if is_cc else 0.5)))
        self.frame_window = (1.0 / floor + self.dt_window) if is_cc else self.dt_window

    def grid(self, n, x1=0.5 * DX):
        if self.half_window < 2:
            return 0, 0.0
        return short_term_frames(n, self.frame_window, self.dt, x1)

    def tables(self):
        if self.is_cc:
            return None, None
        n = self.nsamp_window
        i = np.arange(1, n + 1)
        win = 0.5 - 0.5 * np.cos(i * 2.0 * np.pi / (n + 1))
        nfft = 1
        while nfft < n * 1.5:
            nfft *= 2
        sp = np.fft.rfft(win, n=nfft)
        wr = np.fft.irfft(sp.real ** 2 + sp.imag ** 2, n=nfft)
        wr = wr / wr[0]
        return win, wr[:self.brent_ixmax + 1].copy()


def lowpass_eligible(lengths):
    """Per clip: does Praat's whole-sound FFT low-pass (the first step of To Formant (burg)) fit the device transform?"""
    cap = int(_lib.load().rsaf_praat_lowpass_max_samples())
    return [int(n) <= cap for n in lengths]


class MshdsEngine:
    def __init__(self, device="cuda"):
        import torch
        _lib.load()
        _lib.require_gpu()
        self.device = torch.device(device)
        self._tables = {}
        self.fo_doubles = _lib.load().rsaf_mshds_frameout_doubles()
        import os
        self.n_streams = int(os.environ.get("RSAF_MSHDS_STREAMS", "3"))     # 1: every analysis on the caller's stream
        self._aux = None

    # ---- cached device tables ----
    def _table(self, key, builder):
        if key not in self._tables:
            self._tables[key] = tuple(None if a is None else _dev(np.asarray(a, dtype=np.float64), self.device, blocking=True)
                                      for a in builder())
        return self._tables[key]

    # ---- one pitch analysis over a set of clips ----
    def pitch(self, wav, sample_offs, lengths, gpeak, *, time_step, floor, ceiling, max_candidates=15,
              silence_threshold=0.03, voicing_threshold=0.45, octave_cost=0.01, octave_jump_cost=0.35,
              voiced_unvoiced_cost=0.14, periods=3.0, is_cc=False, refine_depth=70, voicing_threshold2=None,
              stream=None, dom=None):
        """One Sound: To Pitch (ac/cc) analysis.  ``voicing_threshold2``: also return (key ``second``) the same
        analysis with that voicing threshold; the frame kernel's correlation and refinement are shared."""
        import torch
        lib = _lib.load()
        g = _PitchGeom(time_step, floor, ceiling, periods, is_cc)
        ci, total, mx = _clip_info(sample_offs, lengths, g.grid, dom)
        n = len(lengths)
        dev = self.device
        ci_d = _dev(ci, dev)
        win, wr = self._table(("pitch", g.nsamp_window, is_cc), g.tables)
        tf = max(total, 1)
        frame_out = torch.empty(tf * self.fo_doubles, dtype=torch.float64, device=dev)
        psi = torch.empty(tf * 16, dtype=torch.uint8, device=dev)
        end_state = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        sel_f = torch.zeros(tf, dtype=torch.float64, device=dev)
        sel_s = torch.zeros(tf, dtype=torch.float64, device=dev)
        stats = torch.empty((max(n, 1), 8), dtype=torch.float64, device=dev)
        # cross-correlation passes can have candidates whose interpolation depth the array ends clip: the per-depth tables
        # ride behind the shared one when they are small (depth 70: 618 KB; depth 700 would be 63 MB: direct evaluation)
        clipped_tables = bool(is_cc) and int(refine_depth) <= CHEB_CLIP_MAX_DEPTH
        # deeper cross-correlation passes (the harmonicity pass: depth 700) get one table per cell instead: built once per
        # geometry, 7-19 MB, and used by the per-cell coefficient kernel for clipped and unclipped cells alike
        table_mode = 1 if clipped_tables else 0
        cheb = None
        if is_cc and not clipped_tables and voicing_threshold2 is None and g.half_window >= 2:
            (cell,) = self._table(("sinc_cell", g.max_lag, g.brent_ixmax, g.min_lag, int(refine_depth)),
                                  lambda: (sinc_cell_tables(g.max_lag, g.brent_ixmax, g.min_lag, int(refine_depth)),))
            if cell is not None:
                table_mode, cheb = 2, cell
        params = (C.c_double * 18)(g.dt, g.floor, g.ceiling, voicing_threshold, octave_cost, silence_threshold,
                                   octave_jump_cost, voiced_unvoiced_cost, g.nsamp_window, g.nsamp_period, g.min_lag,
                                   g.max_lag, g.brent_ixmax, max_candidates, refine_depth, 1 if is_cc else 0,
                                   g.dt_window, table_mode)
        second = None
        if voicing_threshold2 is not None:
            second = {"geom": g, "ci": ci, "ci_dev": ci_d, "total_frames": total, "max_frames": mx,
                      "frame_out": torch.empty(tf * self.fo_doubles, dtype=torch.float64, device=dev),
                      "psi": torch.empty(tf * 16, dtype=torch.uint8, device=dev),
                      "end_state": torch.empty(max(n, 1), dtype=torch.int32, device=dev),
                      "sel_freq": torch.zeros(tf, dtype=torch.float64, device=dev),
                      "sel_strength": torch.zeros(tf, dtype=torch.float64, device=dev),
                      "stats": torch.empty((max(n, 1), 8), dtype=torch.float64, device=dev)}
        wp = _lib.ptr(win) if win is not None else None
        wrp = _lib.ptr(wr) if wr is not None else None
        if cheb is None:
            (cheb,) = self._table(("sinc_cheb", int(refine_depth), clipped_tables),
                                  lambda: (sinc_cheb_tables(int(refine_depth), clipped_tables),))
        chp = _lib.ptr(cheb)
        # correlation rows between the two pitch kernels: all clips at once if that stays below ~4 GB, else in groups
        per_clip = int(lib.rsaf_mshds_pitch_workspace_bytes_per_clip(mx, params)) if n else 0
        ws_bytes = max(per_clip * max(1, min(n, int(self.pitch_ws_cap_bytes // max(per_clip, 1)))), 8)
        ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=dev)
        if n and g.half_window >= 2:
            if second is None:
                _lib.check(lib.rsaf_mshds_pitch(
                    _lib.ptr(wav), _lib.ptr(ci_d), n, mx, _lib.ptr(gpeak), wp, wrp, params,
                    _lib.ptr(frame_out), _lib.ptr(psi), _lib.ptr(end_state), _lib.ptr(sel_f), _lib.ptr(sel_s),
                    _lib.ptr(stats), chp, _lib.ptr(ws), ws_bytes, _lib.stream_ptr(stream)), "rsaf_mshds_pitch")
            else:
                _lib.check(lib.rsaf_mshds_pitch_dual(
                    _lib.ptr(wav), _lib.ptr(ci_d), n, mx, _lib.ptr(gpeak), wp, wrp, params,
                    _lib.ptr(frame_out), _lib.ptr(psi), _lib.ptr(end_state), _lib.ptr(sel_f), _lib.ptr(sel_s),
                    _lib.ptr(stats), float(voicing_threshold2), _lib.ptr(second["frame_out"]), _lib.ptr(second["psi"]),
                    _lib.ptr(second["end_state"]), _lib.ptr(second["sel_freq"]), _lib.ptr(second["sel_strength"]),
                    _lib.ptr(second["stats"]), chp, _lib.ptr(ws), ws_bytes, _lib.stream_ptr(stream)), "rsaf_mshds_pitch_dual")
        else:
            stats.fill_(float("nan"))
            stats[:, 0] = 0
            if second is not None:
                second["stats"].fill_(float("nan"))
                second["stats"][:, 0] = 0
        if second is not None:
            second["stats"] = second["stats"][:n]
        return {"geom": g, "ci": ci, "ci_dev": ci_d, "sel_freq": sel_f, "sel_strength": sel_s, "stats": stats[:n],
                "frame_out": frame_out, "total_frames": total, "max_frames": mx, "second": second}

    def intensity(self, wav, sample_offs, lengths, minimum_pitch, time_step, subtract_mean=True, stream=None, dom=None):
        import torch
        lib = _lib.load()
        phys = 6.4 / minimum_pitch
        dt = time_step if time_step > 0 else 0.8 / minimum_pitch
        half_dur = 0.5 * phys
        half = int(math.floor(half_dur / DX))

        def build():
            i = np.arange(-half, half + 1)
            xx = i * DX / half_dur
            return (np.i0((2.0 * np.pi * np.pi + 0.5) * np.sqrt(np.maximum(0.0, 1.0 - xx * xx))),)
        (win,) = self._table(("intensity", half, minimum_pitch), build)
        ci, total, mx = _clip_info(sample_offs, lengths, lambda n, x1: short_term_frames(n, phys, dt, x1), dom)
        n = len(lengths)
        db = torch.empty(max(total, 1), dtype=torch.float64, device=self.device)
        stats = torch.empty((max(n, 1), 2), dtype=torch.float64, device=self.device)
        if n:
            _lib.check(lib.rsaf_mshds_intensity(_lib.ptr(wav), _lib.ptr(_dev(ci, self.device)), n, mx, _lib.ptr(win),
                                                half, dt, 1 if subtract_mean else 0, _lib.ptr(db), _lib.ptr(stats),
                                                _lib.stream_ptr(stream)), "rsaf_mshds_intensity")
        return {"db": db, "ci": ci, "ci_dev": _dev(ci, self.device), "stats": stats[:n], "dt": dt, "max_frames": mx}

    def speechrate(self, wav, sample_offs, lengths, gpeak, stream=None, dom=None):
        """``_speechrate`` (:11-125): intensity(50 Hz, 16 ms) + the 4-candidate pitch pass of :104 ->
        float64 [n, 5].  (The harmonicity call of :36-38 is not evaluated: its value only feeds a no-op and its failure path - clips shorter than 26.7 ms - lies inside the failure path of the intensity call of :41 - clips shorter than 128 ms -, which gives the same five NaN; tests/test_mshds_oracle.py checks the containment.)"""
        import torch
        lib = _lib.load()
        n = len(lengths)
        inten = self.intensity(wav, sample_offs, lengths, 50.0, 0.016, True, stream, dom)            # :41
        p = self.pitch(wav, sample_offs, lengths, gpeak, time_step=0.02, floor=30.0, ceiling=450.0,
                       max_candidates=4, silence_threshold=0.03, voicing_threshold=0.25, octave_cost=0.01,
                       octave_jump_cost=0.35, voiced_unvoiced_cost=0.25, stream=stream, dom=dom)   # :104
        out = torch.empty((max(n, 1), 5), dtype=torch.float64, device=self.device)
        if n:
            wsd = int(lib.rsaf_mshds_speechrate_workspace_doubles(inten["max_frames"]))
            ws = torch.empty(n * wsd, dtype=torch.float64, device=self.device)
            _lib.check(lib.rsaf_mshds_speechrate(_lib.ptr(inten["db"]), _lib.ptr(inten["ci_dev"]), n, inten["max_frames"],
                                                 inten["dt"], _lib.ptr(p["sel_freq"]), _lib.ptr(p["ci_dev"]),
                                                 p["geom"].dt, p["geom"].ceiling, _lib.ptr(ws), _lib.ptr(out),
                                                 _lib.stream_ptr(stream)), "rsaf_mshds_speechrate")
        return out[:n]

    def spectral_moments(self, wav, sample_offs, lengths, pitch, window_length=0.025, time_step=0.005,
                         maximum_frequency=5000.0, frequency_step=20.0, stream=None, dom=None):
        import torch
        lib = _lib.load()
        nyq = 0.5 / DX
        phys = 2.0 * window_length
        eff_t = window_length / math.sqrt(math.pi)
        tstep = max(time_step, eff_t / 8.0)
        fstep = max(frequency_step, (1.0 / eff_t) / 8.0)
        nsamp = int(math.floor(phys / DX))
        half = nsamp // 2 - 1
        nsamp = half * 2
        fmax = maximum_frequency if 0 < maximum_frequency <= nyq else nyq
        nfreq = int(math.floor(fmax / fstep))
        nfft = 1
        while nfft < nsamp or nfft < 2 * nfreq * (nyq / fmax):
            nfft *= 2
        bw_samples = max(1, int(math.floor(fstep * DX * nfft)))
        if bw_samples != 1:
            raise _lib.RsafError("spectrogram: only one FFT bin per frequency band is implemented")
        fstep = 1.0 / (DX * nfft)
        nfreq = int(math.floor(fmax / fstep))

        def build():
            i = np.arange(1, nsamp + 1)
            phase = (i - 0.5 * (nsamp + 1)) / nsamp
            edge = math.exp(-12.0)
            win = (np.exp(-48.0 * phase * phase) - edge) / (1.0 - edge)
            k = np.arange(nfft)
            tw = np.stack([np.cos(2.0 * np.pi * k / nfft), -np.sin(2.0 * np.pi * k / nfft)], axis=1)
            return win, tw.reshape(-1)
        win, tw = self._table(("spec", nsamp, nfft), build)

        def grid(n, x1):
            duration = n * DX
            if phys > duration or half < 1:
                return 0, 0.0
            nt = 1 + int(math.floor((duration - phys) / tstep))
            return nt, x1 + 0.5 * ((n - 1) * DX - (nt - 1) * tstep)
        ci, total, mx = _clip_info(sample_offs, lengths, grid, dom)
        n = len(lengths)
        mom = torch.empty(max(total, 1) * 5, dtype=torch.float64, device=self.device)
        stats = torch.empty((max(n, 1), 4), dtype=torch.float64, device=self.device)
        if n:
            _lib.check(lib.rsaf_mshds_spectral_moments(
                _lib.ptr(wav), _lib.ptr(_dev(ci, self.device)), _lib.ptr(pitch["ci_dev"]), n, mx,
                _lib.ptr(pitch["sel_freq"]), pitch["geom"].dt, pitch["geom"].ceiling, _lib.ptr(win), _lib.ptr(tw),
                nsamp, nfft, nfreq, tstep, fstep, _lib.ptr(mom), _lib.ptr(stats), _lib.stream_ptr(stream)),
                "rsaf_mshds_spectral_moments")
        return {"stats": stats[:n], "moments": mom, "ci": ci, "fstep": fstep, "tstep": tstep}

    def formants(self, wav, sample_offs, lengths, gpeak, floor, ceiling, frame_shift=0.005, stream=None, dom=None):
        """``_measureFormants`` (:303-338) -> float64 [n, 8] (mean/SD of F1, B1, F2, B2 at the pulses)."""
        import torch
        lib = _lib.load()
        n = len(lengths)
        dev = self.device
        dom = dom if dom is not None else SoundDomain(lengths)
        # To Formant (burg) low-passes the WHOLE sound by one FFT (Sound_resample): a clip beyond the transform's limit
        # (2^26 - 2000 samples = 69 min at 16 kHz) loses its eight formant columns (NaN), nothing else, and nobody else's
        ok = lowpass_eligible(lengths)
        if not all(ok):
            out = torch.full((n, 8), float("nan"), dtype=torch.float64, device=dev)
            ids = [i for i in range(n) if ok[i]]
            if ids:
                sub = self.formants(wav, [sample_offs[i] for i in ids], [lengths[i] for i in ids],
                                    gpeak[_dev(np.asarray(ids, dtype=np.int64), dev)].contiguous(), floor, ceiling, frame_shift,
                                    stream, dom.sub(ids))
                out[_dev(np.asarray(ids, dtype=np.int64), dev)] = sub
            return out
        dxo = 1.0 / RS_RATE
        ratio = RS_RATE / FS
        ri = np.zeros(n, dtype=RESAMPLE_INFO)
        lps = np.zeros(n, dtype=LP_SIG)
        tabs, bases, key_to_table = [], [], {}
        out_off = work_off = 0
        for i, (so, nn) in enumerate(zip(sample_offs, lengths)):
            # Sound_resample: round((xmax - xmin) fs) samples on a grid centred in the sound's DOMAIN; output sample j sits at
            # the real input index (x1o + j dxo - x1) / dx
            m = int(math.floor(dom.xmax[i] * RS_RATE + 0.5))
            x1o = 0.5 * (dom.xmax[i] - (m - 1) / RS_RATE)
            pos0 = (x1o - dom.x1[i]) / DX
            if pos0 not in key_to_table:
                key_to_table[pos0] = len(tabs)
                rows, bs = resample10k_tables(pos0)
                tabs.append(rows)
                bases.append(bs)
            ri[i] = (int(so), out_off, pos0, x1o, int(nn), m, key_to_table[pos0], 0)
            lg = max(11, int(nn + 2000 - 1).bit_length())
            lps[i] = (int(so), int(so), work_off, int(nn), lg)
            out_off += m
            work_off += 1 << (lg - 1)
        max_out = int(ri["n_out"].max()) if n else 0
        y10 = torch.empty(max(out_off, 1), dtype=torch.float64, device=dev)
        out = torch.full((max(n, 1), 8), float("nan"), dtype=torch.float64, device=dev)
        if n == 0:
            return out[:0]
        ri_d = _dev(ri, dev)
        wstride = int(lib.rsaf_mshds_resample10k_table_stride(RS_DEPTH))   # rows zero-padded for the kernel's tap blocks
        tab = np.zeros((len(tabs), 5, wstride))
        tab[:, :, :2 * RS_DEPTH + 1] = np.stack(tabs)
        tab_d = _dev(tab.reshape(-1), dev)
        base_d = _dev(np.asarray(bases, dtype=np.int32).reshape(-1), dev)
        # Sound_resample(10000, 500): whole-sound FFT low-pass (16 kHz -> 10 kHz goes down), then sinc interpolation
        lp = torch.empty(int(wav.numel()), dtype=torch.float64, device=dev)
        work = torch.empty(2 * work_off, dtype=torch.float64, device=dev)
        _lib.check(lib.rsaf_praat_lowpass_batch(_lib.ptr(wav), _lib.ptr(_dev(lps, dev)), n, int(lps["lg"].max()), RS_RATE * DX,
                                                _lib.ptr(work), work_off, _lib.ptr(lp), _lib.stream_ptr(stream)),
                   "rsaf_praat_lowpass_batch")
        _lib.check(lib.rsaf_mshds_resample10k(_lib.ptr(lp), _lib.ptr(ri_d), n, max_out, _lib.ptr(tab_d), wstride, _lib.ptr(base_d),
                                              RS_DEPTH, _lib.ptr(y10), _lib.stream_ptr(stream)), "rsaf_mshds_resample10k")
        del work
        # Formant (burg): 5 ms, 5 formants, 5 kHz, 25 ms half-window, pre-emphasis from 50 Hz  (:319)
        dt_window = 0.05
        nsw = int(math.floor(dt_window / dxo))

        def build():
            i = np.arange(1, nsw + 1)
            imid, edge = 0.5 * (nsw + 1), math.exp(-12.0)
            return ((np.exp(-48.0 * (i - imid) ** 2 / (nsw + 1) ** 2) - edge) / (1.0 - edge),)
        (win,) = self._table(("formant", nsw), build)
        ci = np.zeros(n, dtype=CLIP_INFO)
        foff = 0
        for i in range(n):
            m, x1o = int(ri[i]["n_out"]), float(ri[i]["x1o"])
            duration = m * dxo
            if dt_window > duration:
                nf, t1 = 0, 0.0
            else:
                nf = int(math.floor((duration - dt_window) / frame_shift)) + 1
                t1 = x1o - 0.5 * dxo + 0.5 * duration - 0.5 * nf * frame_shift + 0.5 * frame_shift
            ci[i] = (int(ri[i]["out_off"]), foff, t1, m, nf, x1o, dom.xmax[i])
            foff += nf
        mxf = int(ci["n_frames"].max())
        frames = torch.empty(max(foff, 1) * 10, dtype=torch.float64, device=dev)
        ci_d = _dev(ci, dev)
        _lib.check(lib.rsaf_mshds_formants(_lib.ptr(y10), _lib.ptr(ri_d), _lib.ptr(ci_d), n, mxf, _lib.ptr(win), nsw,
                                           frame_shift, dxo, math.exp(-2.0 * math.pi * 50.0 * dxo), _lib.ptr(frames),
                                           _lib.stream_ptr(stream)), "rsaf_mshds_formants")
        # To Pitch (cc) with parselmouth's defaults (:320) and the pulses (:321)
        p = self.pitch(wav, sample_offs, lengths, gpeak, time_step=frame_shift, floor=floor, ceiling=ceiling,
                       periods=1.0, is_cc=True, refine_depth=70, stream=stream, dom=dom)
        pulses, npul, max_pulses = self.pulses(wav, lengths, p, stream)
        _lib.check(lib.rsaf_mshds_formant_stats(_lib.ptr(frames), _lib.ptr(ci_d), n, frame_shift, _lib.ptr(pulses),
                                                max_pulses, _lib.ptr(npul), _lib.ptr(out), _lib.stream_ptr(stream)),
                   "rsaf_mshds_formant_stats")
        self._last_formants = {"frames": frames, "ci": ci, "pulses": pulses, "n_pulses": npul, "max_pulses": max_pulses,
                               "y10": y10, "ri": ri, "pitch": p}
        return out[:n]

    def pulses(self, wav, lengths, pitch, stream=None):
        """Sound & Pitch: To PointProcess (cc): pulse times per clip in ascending order -> (pulses, n_pulses, max_pulses)."""
        import torch
        n = len(lengths)
        ceiling = pitch["geom"].ceiling
        max_pulses = int(max(lengths) * DX * ceiling * 1.5) + 16
        pulses = torch.empty(max(n, 1) * max_pulses, dtype=torch.float64, device=self.device)
        npul = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        if n:
            lib = _lib.load()
            mx = int(pitch["max_frames"])
            need = int(lib.rsaf_mshds_pulses_workspace_bytes(n, mx, pitch["geom"].dt, ceiling))
            ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
            _lib.check(lib.rsaf_mshds_pulses(_lib.ptr(wav), _lib.ptr(pitch["ci_dev"]), n, mx, _lib.ptr(pitch["sel_freq"]),
                                             pitch["geom"].dt, ceiling, _lib.ptr(ws), need, _lib.ptr(pulses), max_pulses,
                                             _lib.ptr(npul), _lib.stream_ptr(stream)), "rsaf_mshds_pulses")
        return pulses, npul, max_pulses

    def slope_tilt(self, wav, sample_offs, lengths, gpeak, floor, ceiling, stream=None, dom=None):
        """_extract_Slope_Tilt (src/mshds_extractor.py:227-251) -> float64 [n, 2] = (Spectral_Slope, Spectral_Tilt).
        "To Ltas (pitch-corrected)" finds its own pulses: To Pitch (ac) with the standard settings and the
        automatic time step 0.75 / floor, then the cc pulse train."""
        import torch
        n = len(lengths)
        out = torch.full((max(n, 1), 2), float("nan"), dtype=torch.float64, device=self.device)
        if n == 0:
            return out[:0]
        p = self.pitch(wav, sample_offs, lengths, gpeak, time_step=0.0, floor=floor, ceiling=ceiling, stream=stream, dom=dom)
        pulses, npul, max_pulses = self.pulses(wav, lengths, p, stream)
        _lib.check(_lib.load().rsaf_mshds_ltas_slope_tilt(_lib.ptr(wav), _lib.ptr(p["ci_dev"]), n, _lib.ptr(pulses),
                                                          max_pulses, _lib.ptr(npul), 0.0001, 0.02, 1.3, _lib.ptr(out),
                                                          _lib.stream_ptr(stream)), "rsaf_mshds_ltas_slope_tilt")
        self._last_ltas = {"pitch": p, "pulses": pulses, "n_pulses": npul, "max_pulses": max_pulses}
        return out[:n]

    pitch_ws_cap_bytes = 16.0e9  # cap of the workspace between the pitch kernels: correlation rows, coefficient blocks, frame records (clips run in groups)
    CPP_CHUNK = 48          # most clips per launch group (the cepstrogram workspace is ~68 MB per 30 s clip)

    def cpp(self, wav, sample_offs, lengths, gpeak, floor, ceiling, frame_shift=0.005, stream=None, pitch=None, dom=None):
        """_extract_CPP (src/mshds_extractor.py:253-301) -> float64 [n] mean CPPS of the voiced intervals."""
        import torch
        lib = _lib.load()
        n = len(lengths)
        dev = self.device
        out = torch.full((max(n, 1),), float("nan"), dtype=torch.float64, device=dev)
        if n == 0:
            return out[:0]
        p = pitch if pitch is not None else self.pitch(wav, sample_offs, lengths, gpeak, time_step=frame_shift, floor=floor,
                                                       ceiling=ceiling, voicing_threshold=0.3, stream=stream, dom=dom)   # :270
        pulses, npul, max_pulses = self.pulses(wav, lengths, p, stream)                                  # :271

        def build_win():
            i = np.arange(1, 1001)
            imid, edge = 0.5 * 1001, math.exp(-12.0)
            return ((np.exp(-48.0 * (i - imid) ** 2 / 1001 ** 2) - edge) / (1.0 - edge),)

        def build_tw():
            k = np.arange(512)
            return (np.stack([np.cos(2.0 * np.pi * k / 1024.0), -np.sin(2.0 * np.pi * k / 1024.0)], axis=1).reshape(-1),)
        (win,) = self._table(("cpp_window", 1000), build_win)
        (tw,) = self._table(("cpp_twiddle", 1024), build_tw)
        dur = max(lengths) * DX
        max_seg = int(dur / 0.02) + 2
        cap_res = int(dur * 10000.0) + max_seg + 16
        cap_frames = int(dur / 0.002) + max_seg
        # Sound_resample of every extracted part: transforms of the first power of two >= part + 2000 samples, i.e. fewer
        # than part + 2000 complex numbers each
        cap_work = int(max(lengths)) + 2000 * max_seg
        # (a voiced interval longer than the transform's limit - 69 min of unbroken voicing - fails that clip alone: NaN)
        lg_max = min(26, max(11, int(max(lengths) + 2000 - 1).bit_length()))
        segd = int(lib.rsaf_mshds_cpp_seg_doubles())
        ci_all = p["ci"]
        # clips per launch group: bound the cepstrogram workspace (cap_frames x 513 doubles per clip) to ~3 GB
        chunk = max(1, min(self.CPP_CHUNK, int(3.0e9 // (cap_frames * 513 * 8))))
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            m = c1 - c0
            ci_d = _dev(ci_all[c0:c1], dev)
            segs = torch.empty(m * max_seg * segd, dtype=torch.float64, device=dev)
            hdr = torch.zeros(m * 4, dtype=torch.int32, device=dev)
            res = torch.empty(m * cap_res, dtype=torch.float64, device=dev)
            ceps = torch.empty(m * cap_frames * 513, dtype=torch.float64, device=dev)
            cppf = torch.empty(m * cap_frames, dtype=torch.float64, device=dev)
            lp_origin = int(min(int(r["sample_off"]) for r in ci_all[c0:c1]))
            lp_end = int(max(int(r["sample_off"]) + int(r["n_samples"]) for r in ci_all[c0:c1]))
            lp = torch.empty(max(lp_end - lp_origin, 1), dtype=torch.float64, device=dev)
            lpw = torch.empty(m * cap_work * 2, dtype=torch.float64, device=dev)
            _lib.check(lib.rsaf_mshds_cpp(_lib.ptr(wav), _lib.ptr(ci_d), m, _lib.c_void_p_off(pulses, c0 * max_pulses),
                                          max_pulses, _lib.c_void_p_off(npul, c0), _lib.ptr(win), _lib.ptr(tw), max_seg,
                                          cap_res, cap_frames, _lib.ptr(segs), _lib.ptr(hdr), _lib.ptr(res), _lib.ptr(ceps),
                                          _lib.ptr(cppf), _lib.ptr(lp), lp_origin, _lib.ptr(lpw), cap_work, lg_max,
                                          _lib.c_void_p_off(out, c0), _lib.stream_ptr(stream)),
                       "rsaf_mshds_cpp")
            self._last_cpp = {"segs": segs, "hdr": hdr, "res": res, "ceps": ceps, "cpp_frames": cppf, "max_seg": max_seg,
                              "cap_res": cap_res, "cap_frames": cap_frames, "seg_doubles": segd, "pulses": pulses,
                              "n_pulses": npul, "max_pulses": max_pulses, "chunk": (c0, c1)}
        return out[:n]

    def hnr_mean(self, pitch_cc, stream=None):
        import torch
        n = len(pitch_cc["ci"])
        out = torch.empty(max(n, 1), dtype=torch.float64, device=self.device)
        if n:
            _lib.check(_lib.load().rsaf_mshds_hnr_mean(_lib.ptr(pitch_cc["sel_freq"]), _lib.ptr(pitch_cc["sel_strength"]),
                                                       _lib.ptr(pitch_cc["ci_dev"]), n, _lib.ptr(out),
                                                       _lib.stream_ptr(stream)), "rsaf_mshds_hnr_mean")
        return out[:n]

    def clip_peaks(self, wav, sample_offs, lengths, stream=None):
        import torch
        n = len(lengths)
        ci, _, _ = _clip_info(sample_offs, lengths, lambda k, x1: (0, 0.0))
        gp = torch.zeros(max(n, 1), dtype=torch.float64, device=self.device)
        if n:
            _lib.check(_lib.load().rsaf_mshds_clip_peak(_lib.ptr(wav), _lib.ptr(_dev(ci, self.device)), n, _lib.ptr(gp),
                                                        _lib.stream_ptr(stream)), "rsaf_mshds_clip_peak")
        return gp

    # ---- the reference's orchestration for a packed batch ----
    def extract_packed(self, wav, sample_offs, lengths, stream=None, x1=None, xmax=None, before_host_sync=None):
        """wav: 1-D float32 device tensor with the clips back to back -> (float64 [n, 25] device tensor,
        list of (floor, ceiling) per clip).  ``x1`` / ``xmax`` (per clip, seconds): time of the first sample and end of the
        time domain of each sound as Praat carries them (``SoundDomain``); default = sounds read from 16 kHz files.
        ``before_host_sync``: called once, right before the host reads the speaker ranges back (the one point where the host
        waits for the device): whatever it queues keeps the GPU busy while the host decides the ranges and queues the rest."""
        import torch
        n = len(lengths)
        out = torch.full((n, 25), float("nan"), dtype=torch.float64, device=self.device)
        if n == 0:
            return out, []
        sample_offs = [int(v) for v in sample_offs]
        lengths = [int(v) for v in lengths]
        dom = SoundDomain(lengths, x1, xmax)
        gpeak = self.clip_peaks(wav, sample_offs, lengths, stream)
        # Two HIP streams when the caller leaves the stream choice to us: the analyses of a clip are independent once the
        # speaker range is known, and several of them are latency-bound (path finder, pulse walk, speech-rate scan: one
        # wave per clip or stretch), so they run beside the correlation-heavy passes instead of in front of them.
        two = stream is None and self.n_streams > 1
        main = torch.cuda.current_stream(self.device) if two else None
        if two and self._aux is None:
            self._aux = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        three = two and self.n_streams > 2

        def side(fn, which=0):
            """Run fn() on an auxiliary stream (after everything queued on the main stream so far)."""
            if not two:
                return fn()
            aux = self._aux[which if three else 0]
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                return fn()

        def join(*tensors):
            if two:
                for aux in self._aux[:2 if three else 1]:
                    main.wait_stream(aux)
                for t in tensors:
                    t.record_stream(main)

        sr = side(lambda: self.speechrate(wav, sample_offs, lengths, gpeak, stream, dom))           # :426
        # _pitch_values (:127-162): wide search, outlier-trimmed mean -> speaker range
        wide = self.pitch(wav, sample_offs, lengths, gpeak, time_step=0.005, floor=50.0, ceiling=600.0, stream=stream, dom=dom)
        if before_host_sync is not None:
            before_host_sync()
        st = wide["stats"].cpu().numpy()                       # one small D2H per batch
        join(sr)
        out[:, 0:5] = sr
        ranges = []
        for i in range(n):
            if st[i, 0] == 0 or not st[i, 7] > 0:
                ranges.append((75, 500))                       # :146,151 and the except path :161-162
            else:
                ranges.append((60, 250) if st[i, 3] < 170 else (100, 500))
        for rng in sorted(set(ranges)):
            ids = [i for i in range(n) if ranges[i] == rng]
            so = [sample_offs[i] for i in ids]
            ln = [lengths[i] for i in ids]
            dm = dom.sub(ids)
            idx = _dev(np.asarray(ids, dtype=np.int64), self.device)
            gp = gpeak[idx].contiguous()
            floor, ceiling = float(rng[0]), float(rng[1])

            def branch_hnr():
                inten = self.intensity(wav, so, ln, floor, 0.005, True, stream, dm)                          # :198
                cc = self.pitch(wav, so, ln, gp, time_step=0.005, floor=floor, ceiling=0.5 / DX, max_candidates=15,
                                silence_threshold=0.1, voicing_threshold=0.0, octave_cost=0.0, octave_jump_cost=0.0,
                                voiced_unvoiced_cost=0.0, periods=4.5, is_cc=True, refine_depth=700, stream=stream, dom=dm)  # :221
                return inten["stats"], self.hnr_mean(cc, stream)

            def branch_pulses():
                fm = self.formants(wav, so, ln, gp, floor, ceiling, 0.005, stream, dm)                       # :441
                return fm, self.slope_tilt(wav, so, ln, gp, floor, ceiling, stream, dm)                      # :433

            i_stats, hnr = side(branch_hnr, 0)
            fm, sl = side(branch_pulses, 1)
            # :178 == :355, and :270 (voicing threshold 0.3, everything else equal) from the same frame kernel
            p = self.pitch(wav, so, ln, gp, time_step=0.005, floor=floor, ceiling=ceiling, voicing_threshold2=0.3,
                           stream=stream, dom=dm)
            sm = self.spectral_moments(wav, so, ln, p, 0.025, 0.005, stream=stream, dom=dm)                  # :356
            cppv = self.cpp(wav, so, ln, gp, floor, ceiling, 0.005, stream, pitch=p["second"], dom=dm)       # :434
            join(i_stats, hnr, fm, sl)
            out[idx, 5] = p["stats"][:, 5]
            out[idx, 6] = p["stats"][:, 6]
            out[idx, 7] = i_stats[:, 0]
            out[idx, 8] = i_stats[:, 1]
            out[idx, 9] = hnr
            out[idx, 10:12] = sl
            out[idx, 12] = cppv
            out[idx, 13:21] = fm
            out[idx, 21:25] = sm["stats"]
        return out, ranges


_ENGINE = None


def get_engine(device="cuda"):
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = MshdsEngine(device)
    return _ENGINE


def extract_mshds_features(input_df, audio_file_column="filepath", verbose=True, batch_files=64):
    """Drop-in for ``src/mshds_extractor.py:379-459``: one row per input row, in input order,
    columns ``filename`` + the 25 feature names; a file that cannot be processed gives a NaN row
    (``:450-457``).  Files at another sample rate are converted to 16 kHz on the device first."""
    import pandas as pd
    import torch
    eng = get_engine()
    rows = []
    paths = list(input_df[audio_file_column])
    for b0 in range(0, len(paths), batch_files):
        batch = paths[b0:b0 + batch_files]
        clips, ok_idx, x1s, xmaxs = [], [], [], []
        for j, pth in enumerate(batch):
            filename = os.path.basename(pth)
            try:
                x, fs, n_in = read_wav_mono_device(pth, device=eng.device)     # :415-416 Sound(path), convert_to_mono
                x1, xmax = 0.5 / fs, int(x.numel()) / fs                      # a Sound read from a file
                if fs != SAMPLE_RATE:                                         # :418-419 snd.resample(16000, 50)
                    from .resample import resample_praat_sound
                    x, x1, xmax = resample_praat_sound(x, fs, SAMPLE_RATE, 50, device=eng.device)
                if int(x.numel()) == 0:
                    raise ValueError("empty file")
                clips.append(x)
                ok_idx.append(j)
                x1s.append(x1)
                xmaxs.append(xmax)
            except Exception as e:
                if verbose:
                    print(f"ERROR processing file '{filename}': {e}. Appending NaNs.")
        feats = np.full((len(batch), 25), np.nan)
        if clips:
            lengths = [int(c.numel()) for c in clips]
            offs = np.zeros(len(clips) + 1, dtype=np.int64)
            offs[1:] = np.cumsum(lengths)
            wav = torch.cat(clips) if len(clips) > 1 else clips[0].contiguous()
            try:
                vals, _ = eng.extract_packed(wav, offs[:-1], lengths, x1=x1s, xmax=xmaxs)
                torch.cuda.synchronize()
                feats[ok_idx] = vals.cpu().numpy()
            except _lib.RsafError as e:
                # a clip the kernels reject must not take its batch mates down: redo the batch clip by clip and give
                # only the offending files the reference's per-file NaN row (:450-457)
                if verbose:
                    print(f"WARNING: batch of {len(clips)} files failed ({e}); retrying file by file.")
                for k, (j, c) in enumerate(zip(ok_idx, clips)):
                    try:
                        v1, _ = eng.extract_packed(c.contiguous(), [0], [int(c.numel())], x1=x1s[k:k + 1], xmax=xmaxs[k:k + 1])
                        torch.cuda.synchronize()
                        feats[j] = v1.cpu().numpy()[0]
                    except _lib.RsafError as e1:
                        if verbose:
                            print(f"ERROR processing file '{os.path.basename(batch[j])}': {e1}. Appending NaNs.")
        for j, pth in enumerate(batch):
            d = {"filename": os.path.basename(pth)}
            d.update({nme: feats[j, k] for k, nme in enumerate(FEATURE_NAMES)})
            rows.append(d)
    return pd.DataFrame(rows)
