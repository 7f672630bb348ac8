"""Analytic known answers pushed through BOTH sides - the CPU restatements under oracle/ and the HIP kernels - for the parts
of the path whose oracles nothing in the reference can pin (no Praat / SMILExtract here): each test builds a
signal whose answer follows from its construction, states that answer, and checks the oracle AND the device against it
(and, where the quantity is defined per frame, the device against the oracle at full length).

  * Burg formants of a stationary five-resonance all-pole process            (src/mshds_extractor.py:303-338)
  * cepstral peak of an impulse train: quefrency of the maximum = the period (src/mshds_extractor.py:253-301)
  * pitch-corrected Ltas slope of harmonic complexes with prescribed tilts   (src/mshds_extractor.py:227-251)
  * MFCC of a frame with a flat magnitude spectrum                           (Androids.conf:101-115)
  * spectral centroid / variance / roll-off of a two-line spectrum           (Androids.conf:258-280)
  * the whole per-frame pitch track of one 30 s clip                         (src/mshds_extractor.py:143,178)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mshds_oracle as mo
from oracle import smile_oracle as so
from robust_speech_analysis_framework_amd import synth

FS = 16000


def _pack(clips):
    import torch
    lengths = [len(c) for c in clips]
    offs = np.concatenate([[0], np.cumsum(lengths)])[:-1]
    wav = torch.from_numpy(np.concatenate(clips).astype(np.float32)).cuda()
    return wav, [int(o) for o in offs], lengths


@pytest.fixture(scope="module")
def eng(rsaf_lib):
    from robust_speech_analysis_framework_amd.mshds import MshdsEngine
    return MshdsEngine()


def _resonator_cascade(exc, freqs, bws, fs):
    """exc filtered by a cascade of two-pole resonators (poles r e^{+-i theta}, r = exp(-pi bw / fs), theta = 2 pi f / fs)."""
    from scipy.signal import lfilter
    y = np.asarray(exc, dtype=np.float64)
    for f, bw in zip(freqs, bws):
        r, th = np.exp(-np.pi * bw / fs), 2.0 * np.pi * f / fs
        y = lfilter([1.0], [1.0, -2.0 * r * np.cos(th), r * r], y)
    return y


def test_burg_formants_of_a_five_resonance_all_pole_process(eng):
    """White noise through five narrow resonators = an order-10 all-pole process: Burg order 10 (5 formants below 5 kHz on the
    10 kHz resampled sound) must put its five pole pairs on the resonances.  Known answer: F2..F4 = the designed centre
    frequencies within 1.5 %; F1 and F5 within 8 % (the 50 Hz pre-emphasis pulls the lowest pole up, the 5 kHz band edge of
    the resampled sound pulls the highest one down: measured +6 % / -3 % on the oracle).  Oracle and device agree per frame."""
    import torch
    F = np.array([500.0, 1500.0, 2500.0, 3400.0, 4100.0])
    B = [30.0, 30.0, 40.0, 40.0, 50.0]
    bar = np.array([0.08, 0.015, 0.015, 0.015, 0.08]) * F
    rng = np.random.default_rng(11)
    x = _resonator_cascade(rng.standard_normal(int(1.2 * FS)), F, B, FS)
    x = (0.3 * x / np.abs(x).max()).astype(np.float32)
    Fo, Bo, t1, dt = mo.formant_burg(x)                                       # oracle: per-frame formants
    med_o = np.nanmedian(Fo, axis=0)
    assert np.all(np.abs(med_o - F) <= bar), med_o
    wav, offs, lens = _pack([x])
    gp = eng.clip_peaks(wav, offs, lens)
    eng.formants(wav, offs, lens, gp, 100.0, 500.0, 0.005)
    torch.cuda.synchronize()
    L = eng._last_formants
    fr = L["frames"].cpu().numpy().reshape(-1, 10)
    ci = L["ci"][0]
    g = fr[ci["frame_off"]:ci["frame_off"] + ci["n_frames"], :5]
    med_g = np.nanmedian(g, axis=0)
    assert np.all(np.abs(med_g - F) <= bar), med_g                              # the device finds the designed poles
    assert g.shape == Fo.shape and np.array_equal(np.isnan(g), np.isnan(Fo))
    ok = ~np.isnan(Fo)
    assert np.abs(g[ok] - Fo[ok]).max() < 1e-2                                   # and agrees with the oracle frame by frame


def test_cepstral_peak_of_a_pulse_train_sits_at_the_period(eng):
    """A glottal-like pulse train with period T0 = 8 ms (125 Hz) through a vowel-like resonator: the power cepstrum of every
    voiced frame has its maximum (over the quefrency range CPPS searches, 1/330 .. 1/60 s) at T0 = bin 80 of the 0.1 ms
    quefrency axis.  Known answer: argmax = 80 +- 1 on both sides; a noise clip's CPPS lies far below the pulse train's."""
    import torch
    n = int(1.5 * FS)
    exc = np.zeros(n)
    exc[::128] = 1.0                                                           # 128 samples = 8 ms
    x = _resonator_cascade(exc, [700.0, 1200.0], [130.0, 160.0], FS)
    x = (0.4 * x / np.abs(x).max()).astype(np.float32)
    rng = np.random.default_rng(5)
    noise = (0.05 * rng.standard_normal(n)).astype(np.float32)
    wav, offs, lens = _pack([x, noise])
    gp = eng.clip_peaks(wav, offs, lens)
    got = eng.cpp(wav, offs, lens, gp, 100.0, 500.0).cpu().numpy()
    torch.cuda.synchronize()
    L = eng._last_cpp
    hdr = L["hdr"].cpu().numpy().reshape(2, 4)
    segs = L["segs"].cpu().numpy().reshape(2, L["max_seg"], L["seg_doubles"])
    ceps = L["ceps"].cpu().numpy().reshape(2, L["cap_frames"], 513)
    assert hdr[0, 0] >= 1                                                      # the pulse train is voiced
    lo, hi = int(np.ceil(1e4 / 330.0)), int(np.floor(1e4 / 60.0))
    peaks = []
    for k in range(int(hdr[0, 0])):
        f0, nf = int(segs[0, k, 4]), int(segs[0, k, 5])
        for fr in ceps[0, f0:f0 + nf]:
            peaks.append(lo + int(np.argmax(fr[lo:hi + 1])))
    peaks = np.asarray(peaks)
    assert len(peaks) > 50 and np.mean(np.abs(peaks - 80) <= 1) > 0.95, np.bincount(peaks)[70:90]
    ref = mo.extract_cpp(x, 100.0, 500.0)                                      # the oracle's CPPS of the same clip
    assert abs(got[0] - ref) <= 1e-6 * abs(ref)
    assert got[0] > 15.0 and (np.isnan(got[1]) or got[1] < got[0] - 8.0), got   # a clear cepstral peak against none


def test_ltas_slope_of_harmonic_complexes_with_prescribed_tilt(eng):
    """Harmonics of 125 Hz with amplitudes k^-p.  What the construction fixes about Praat's "Get slope 0 1000 1000 4000 energy"
    on the pitch-corrected Ltas (a difference of band levels in dB): (1) it does not change when the sound is halved
    (every bin drops by the same 6.02 dB) - exactly; (2) every harmonic's level relative to the first is -20 p log10 k dB,
    linear in p, so the band-level difference scales with p: slope(p = 2) = 2 slope(p = 1) within 3 %, and it is negative and
    strictly decreasing in p.  Both hold for the oracle and for the device, which also agree with each other to 1e-6."""
    import torch
    n = int(2.0 * FS)
    t = np.arange(n) / FS
    f0 = 125.0
    ks = np.arange(1, 56)
    clips = []
    for p in (1.0, 1.5, 2.0):
        x = sum((float(k) ** -p) * np.sin(2 * np.pi * k * f0 * t + 0.3 * k) for k in ks if k * f0 < 7000)
        x = (0.3 * x / np.abs(x).max()).astype(np.float32)
        clips += [x, (0.5 * x).astype(np.float32)]
    ref = np.array([mo.extract_slope_tilt(c, 100.0, 500.0) for c in clips])
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    got = eng.slope_tilt(wav, offs, lens, gp, 100.0, 500.0).cpu().numpy()
    torch.cuda.synchronize()
    for rows, name in ((ref, "oracle"), (got, "device")):
        s = rows[:, 0]
        assert np.all(np.abs(s[0::2] - s[1::2]) <= 1e-9 * np.abs(s[0::2])), (name, s)        # (1) amplitude invariance
        assert s[0] < 0 and s[0] > s[2] > s[4], (name, s)                                    # (2) steeper tilt, lower slope
        assert abs(s[4] / s[0] - 2.0) < 0.06, (name, s[4] / s[0])
    assert np.all(np.abs(got[:, 0] - ref[:, 0]) <= 1e-6 * np.abs(ref[:, 0]) + 1e-9)
    assert np.all(np.abs(got[:, 1] - ref[:, 1]) <= 1e-6 * np.abs(ref[:, 1]) + 1e-12)


def _smile_rows(clips):
    import torch
    from robust_speech_analysis_framework_amd import smile
    p = smile.pack_clips(clips)
    lld = smile.smile_lld(p)
    torch.cuda.synchronize()
    return lld.cpu().numpy(), p


def test_mfcc_of_a_frame_with_a_flat_magnitude_spectrum(rsaf_lib):
    """One unit impulse per 10 ms hop, placed so that every 25 ms frame holds it at the same in-frame position n0 away from
    the frame start ... the pre-emphasised, Hamming-weighted frame is w[n0] delta[n - n0] - 0.97 w[n0 + 1] delta[n - n0 - 1], whose
    magnitude spectrum is known in closed form: |X[k]| = |w0 - 0.97 w1 e^{-i 2 pi k / N}|.  Known answer: MFCC 1..12 =
    lifted DCT-II of the log of that spectrum through the 26-band HTK mel bank, computed here from the closed form."""
    P = so.Params(16000)
    n = 16000
    x = np.zeros(n, np.float32)
    x[200::160] = 0.5                                                          # hop 160: in-frame position 200 - 160 j ... one per frame
    # frame j covers samples [160 j, 160 j + 400): impulses at 200 + 160 m -> in-frame positions 200 - 160 (j - m): 200 and 40 (and 360)
    got, p = _smile_rows([x])
    ref = so.lld(x)
    j = 10                                                                      # an interior frame
    pos = [q for q in (40, 200, 360)]                                           # in-frame positions of the impulses of every frame
    w = P.hamming()
    k = np.arange(P.nbins)
    X = np.zeros(P.nbins, dtype=np.complex128)
    for q in pos:
        X += 0.5 * w[q] * np.exp(-2j * np.pi * k * q / P.nfft)
        if q + 1 < P.frame:
            X += -so.PREEMPH * 0.5 * w[q + 1] * np.exp(-2j * np.pi * k * (q + 1) / P.nfft)
    mag = np.abs(X)
    melspec = (mag * so.HTK_SCALE) @ P.mel_matrix().T
    want = np.log(np.maximum(melspec, so.MEL_FLOOR)) @ so.dct_matrix().T
    assert np.abs(ref[1:13, j] - want).max() <= 1e-9 * np.abs(want).max()     # the oracle reproduces the closed form
    assert np.abs(got[1:13, j] - want).max() <= 1e-9 * np.abs(want).max()     # and so does the device
    assert np.abs(got[1:13] - ref[1:13]).max() <= 1e-9 * np.abs(ref[1:13]).max()


def test_spectral_descriptors_of_a_two_line_spectrum(rsaf_lib):
    """Two sinusoids on bin centres (bins 32 and 96 of the 512-point transform: 1 000 and 3 000 Hz) with amplitudes 1 and 0.5.
    Without pre-emphasis the power-weighted centroid would be (1000 + 0.25 * 3000) / 1.25 = 1 400 Hz; the chain's pre-emphasis
    |1 - 0.97 e^{-i w}|^2 weights the two lines by g1 = 0.0918..., g3 = 0.6977..., so the known answer is
    centroid = (1000 g1 + 3000 * 0.25 g3) / (g1 + 0.25 g3), variance likewise, 25 % roll-off at the lower line's bin group
    and 90 % at the upper one's.  Hamming leakage spreads each line over +-2 bins symmetrically: 0.5 % bar."""
    n = 8000
    t = np.arange(n) / 16000.0
    f1, f3 = 1000.0, 3000.0
    x = (0.4 * (np.sin(2 * np.pi * f1 * t) + 0.5 * np.sin(2 * np.pi * f3 * t + 0.7))).astype(np.float32)
    g = lambda f: abs(1.0 - so.PREEMPH * np.exp(-2j * np.pi * f / 16000.0)) ** 2      # noqa: E731
    p1, p3 = 1.0 * g(f1), 0.25 * g(f3)
    cen = (f1 * p1 + f3 * p3) / (p1 + p3)
    var = ((f1 - cen) ** 2 * p1 + (f3 - cen) ** 2 * p3) / (p1 + p3)
    got, _ = _smile_rows([x])
    ref = so.lld(x)
    j = slice(2, -2)
    for rows, name in ((ref, "oracle"), (got, "device")):
        assert np.abs(rows[29, j] - cen).max() <= 0.005 * cen, (name, rows[29, 5], cen)
        assert np.abs(rows[31, j] - var).max() <= 0.02 * var, (name, rows[31, 5], var)
        share1 = p1 / (p1 + p3)                                                  # power share of the lower line
        assert share1 < 0.25 or np.all(np.abs(rows[24, j] - f1) <= 2 * 31.25)   # roll-off 25 %
        assert np.all(np.abs(rows[27, j] - f3) <= 2 * 31.25), (name, rows[27, 5])    # roll-off 90 % sits on the upper line
    assert np.abs(got[29] - ref[29]).max() <= 1e-9 * cen and np.array_equal(got[24:28], ref[24:28])   # roll-off bins exact


def test_full_length_pitch_track_of_a_30s_clip_matches_the_oracle_frame_by_frame(eng):
    """BASELINE config C2's unit of work: the wide-range AC analysis of one 30 s clip (5 989 frames): every frame's selected
    frequency and strength against the oracle, voicing decisions identical (the end features alone were compared at this
    length before)."""
    import torch
    clip = synth.synth_clip(20260777, 30.0)
    wav, offs, lens = _pack([clip])
    gp = eng.clip_peaks(wav, offs, lens)
    r = eng.pitch(wav, offs, lens, gp, time_step=0.005, floor=50.0, ceiling=600.0)
    torch.cuda.synchronize()
    p = mo.pitch_ac(clip, 0.005, 50.0, pitch_ceiling=600.0)
    f_ref = p.frequency()
    ci = r["ci"][0]
    assert ci["n_frames"] == len(f_ref) == 5989
    f_got = r["sel_freq"].cpu().numpy()[ci["frame_off"]:ci["frame_off"] + ci["n_frames"]]
    voiced_ref, voiced_got = f_ref > 0, (f_got > 0) & (f_got < 600.0)
    assert np.array_equal(voiced_ref, voiced_got), int(np.sum(voiced_ref != voiced_got))
    assert voiced_ref.sum() > 1000
    assert np.abs(f_got[voiced_ref] - f_ref[voiced_ref]).max() <= 1e-7 * 600.0
