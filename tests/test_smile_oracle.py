"""Analytic known-answer vectors for the openSMILE-style oracle (parity unpinned: no SMILExtract
binary exists here, so the oracle is pinned by closed-form cases only)."""
import numpy as np

from oracle import smile_oracle as so


def test_frame_count_contract():
    assert [so.n_frames(n) for n in (0, 399, 400, 559, 560, 80000, 480000)] == [0, 0, 1, 1, 2, 498, 2998]


def test_feature_names_912_and_order():
    names = so.feature_names()
    assert len(names) == 912 == len(set(names))
    assert names[0] == "pcm_RMSenergy_sma_max"
    assert names[12] == "mfcc_sma[1]_max"
    assert names[16 * 12] == "pcm_RMSenergy_sma_de_max"
    assert names[2 * 16 * 12] == "pcm_intensity_sma_max"
    assert names[-1] == "pcm_fftMag_spectralFlatness_sma_de_kurtosis"


def test_hamming_and_mel_tables():
    h = so.hamming()
    assert abs(h[0] - 0.08) < 1e-12 and abs(h[-1] - 0.08) < 1e-12 and abs(h.max() - 1.0) < 1e-4
    W = so.mel_matrix()
    assert W.shape == (26, 257)
    # interior bins are split between two adjacent triangles whose weights sum to one
    s = W.sum(axis=0)
    lo_chan, _ = so.mel_tables()
    interior = (lo_chan >= 1) & (lo_chan <= 25)
    assert np.allclose(s[interior], 1.0)
    assert (W >= 0).all() and s[0] == 0.0


def test_pure_sine_lands_in_its_bin():
    n = 400 + 160 * 9
    t = np.arange(n) / 16000.0
    x = (0.25 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    L = so.lld(x)
    assert L.shape == (38, 10)
    cen = L[29]
    assert np.all(np.abs(cen - 1000.0) < 40.0)         # centroid at the tone
    assert np.all(L[24:28] <= 1031.25 + 1e-9) and np.all(L[24:28] >= 968.75 - 1e-9)
    zc = L[13]                                           # 1 kHz at 16 kHz: 2 crossings / 16 samples
    assert np.all(np.abs(zc - 2 * 1000.0 / 16000.0) < 0.01)


def test_zcr_of_alternating_signal_and_energy_of_zero():
    x = np.tile(np.array([0.5, -0.5], dtype=np.float32), 400)
    L = so.lld(x)
    assert np.allclose(L[13], 399.0 / 400.0)
    z = so.lld(np.zeros(800, dtype=np.float32))
    assert np.allclose(z[0], 0.0) and np.allclose(z[16], 0.0)
    assert np.allclose(z[1:13], 0.0)                     # log floor 1.0 -> log 0 -> all cepstra 0


def test_delta_of_ramp_is_slope_and_sma_keeps_ramp():
    r = np.arange(50, dtype=np.float64)[None, :] * 0.5
    s = so.sma3(r)
    assert np.allclose(s[0, 1:-1], r[0, 1:-1])
    d = so.delta2(r)
    assert np.allclose(d[0, 2:-2], 0.5)


def test_functionals_of_ramp_and_tie_break():
    r = 2.0 * np.arange(100, dtype=np.float64) + 3.0
    f = so.functionals12(r[None, :])[0]
    assert f[0] == r[-1] and f[1] == r[0] and f[2] == r[-1] - r[0]
    assert f[3] == 99 and f[4] == 0
    assert abs(f[6] - 2.0) < 1e-12 and abs(f[7] - 3.0) < 1e-9 and f[8] < 1e-18
    c = np.array([[1.0, 5.0, 5.0, 0.0, 0.0, 2.0]])
    g = so.functionals12(c)[0]
    assert g[3] == 1 and g[4] == 3                       # first occurrence on ties


def test_extract_shape_and_all_912_finite():
    from robust_speech_analysis_framework_amd import synth
    x = synth.synth_clip(3, 2.0)
    f = so.extract(x)
    assert f.shape == (912,) and np.isfinite(f).all()                  # all 38 LLDs are built


# ---- cFunctionals framing (Androids.conf:349-356) ----------------------------------------------------
def test_functionals_first_window_reading_uses_the_full_length_contours():
    rng = np.random.default_rng(5)
    L = rng.standard_normal((38, 40))
    whole = so.functionals(L)
    first = so.functionals(L, window_frames=3)
    assert whole.shape == first.shape == (912,)
    s = so.sma3(L)[:, :3]
    assert np.isclose(first[0], s[0].max()) and np.isclose(first[5], s[0].mean())
    assert first[3] in (0.0, 1.0, 2.0)                                 # maxPos inside the 3-frame window
    # the delta contour of the window sees frames beyond it (delta regression reaches 2 frames ahead)
    d = so.delta2(so.sma3(L))[:, :3]
    assert np.isclose(first[16 * 12 + 5], d[0].mean())


# ---- native-rate geometry (Androids.conf:73-78: frame sizes are seconds) --------------------------------
def test_params_at_other_sample_rates():
    got = {fs: (so.Params(fs).frame, so.Params(fs).hop, so.Params(fs).nfft) for fs in (8000, 16000, 22050, 44100, 48000)}
    assert got == {8000: (200, 80, 256), 16000: (400, 160, 512), 22050: (551, 221, 1024), 44100: (1103, 441, 2048),
                   48000: (1200, 480, 2048)}
    P = so.Params(8000)
    lo_chan, _ = P.mel_tables()
    assert lo_chan[-1] >= 0 and P.mel_matrix().shape == (26, 129)      # hifreq clipped to Nyquist: top bin still used
    for fs in (8000, 44100):
        t = np.arange(int(0.2 * fs)) / fs
        x = (0.25 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
        L = so.lld(x, so.Params(fs))
        assert L.shape == (38, so.Params(fs).n_frames(len(x))) and np.isfinite(L).all()
        assert np.all(np.abs(L[29] - 1000.0) < 60.0)                   # centroid at the tone at either rate


# ---- cSpecScale ------------------------------------------------------------------------------------------
def test_spec_enhance_keeps_two_bins_around_maxima():
    a = np.array([1, 2, 9, 2, 1, 1, 1, 1, 1, 3, 8, 3, 1, 1], dtype=float)
    e = so.spec_enhance(a)
    #      maxima at 2 and 10: bins 5..7 lie >= 3 from both -> zero; the ends outside the outer maxima are kept
    assert list(e) == [1, 2, 9, 2, 1, 0, 0, 0, 1, 3, 8, 3, 1, 1]
    one = so.spec_enhance(np.array([0, 0, 0, 1, 5, 1, 0.5, 0.4, 0.3, 0.2], dtype=float))
    assert list(one) == [0, 0, 0, 1, 5, 1, 0.5, 0, 0, 0]              # single maximum: all beyond 2 bins zeroed
    flat = np.ones(6)
    assert np.array_equal(so.spec_enhance(flat), flat)                 # no maximum: untouched
    # a[i] >= a[i+1] on the right: the first bin of a plateau is the maximum
    assert list(so.spec_enhance(np.array([0, 1, 4, 4, 3, 2.5, 2, 1.5, 1.2, 2, 0], dtype=float))) == \
        [0, 1, 4, 4, 3, 0, 0, 1.5, 1.2, 2, 0]                          # maxima at 2 (not 3) and 9


def test_spec_smooth_121_with_zero_left_of_the_first_bin():
    a = np.array([4.0, 0.0, 8.0, 4.0])
    assert np.allclose(so.spec_smooth(a), [(0 + 8 + 0) / 4, (4 + 0 + 8) / 4, (0 + 16 + 4) / 4, 4.0])


def test_natural_spline_matches_scipy():
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(1)
    y = rng.random(40)
    cs = CubicSpline(np.arange(40.0), y, bc_type="natural")
    assert np.allclose(so.natural_spline_m(y), cs(np.arange(40.0), 2) / 6.0, atol=1e-12)
    # spec_scale interpolates with exactly that spline (before clipping and weighting)
    P = so.Params(16000)
    mag = rng.random(P.nbins) + 0.5
    a = so.spec_smooth(so.spec_enhance(mag))
    want = np.maximum(CubicSpline(np.arange(P.nbins) * P.df, a, bc_type="natural")(25.0 * 2 ** (P.dl2 * np.arange(P.npts))), 0)
    assert np.allclose(so.spec_scale(mag, P), want * P.auditory_weights(), rtol=1e-9, atol=1e-12)


def test_octave_axis_and_weights():
    P = so.Params(16000)
    f = P.target_pos() * P.df
    assert abs(f[0] - 25.0) < 1e-9 and abs(f[-1] - 8000.0) < 1e-6 and P.npts == 257
    w = P.auditory_weights()
    assert 0.0 < w[0] < 0.5 < w[20] < w[-1] < 1.0 and np.all(np.diff(w) > 0)
    sh = P.shs_shifts()
    assert sh[0] == 0 and sh[1] == int(np.floor(P.ppo)) and len(sh) == 15 and np.all(np.diff(sh) >= 0)


# ---- cPitchShs ------------------------------------------------------------------------------------------
def _tone_mag(f0, harmonics, P, amp=None):
    t = np.arange(P.frame + 4 * P.hop) / P.fs
    x = sum((1.0 if amp is None else amp[h - 1]) * np.sin(2 * np.pi * f0 * h * t) for h in range(1, harmonics + 1))
    x = (0.4 * x / np.abs(x).max()).astype(np.float32)
    return so.magnitudes(x, P)[2]


def test_shs_pure_sine_is_found_at_its_frequency():
    P = so.Params(16000)
    for f0 in (120.0, 200.0, 410.0):
        c = so.shs_candidates(so.spec_scale(_tone_mag(f0, 1, P)[2], P), P)
        assert abs(c[0, 0] - f0) / f0 < 0.01, (f0, c[0])               # one partial: no shift quantisation involved
        assert np.all(c[:, 2][:-1] >= c[:, 2][1:])                      # best score first
        assert np.all((c[:, 0] == 0) | ((c[:, 0] >= 52.0) & (c[:, 0] <= 620.0)))


def test_shs_harmonic_stack_within_the_shift_quantisation_and_voiced():
    """floor(ppo log2 h) moves partial h up to one point (1 / ppo octave = 2.3 % at 16 kHz) to the right: the summed
    peak sits at most that far above f0 (a property of the integer-shift summation itself)."""
    P = so.Params(16000)
    for f0 in (220.0, 330.0):
        c = so.shs_candidates(so.spec_scale(_tone_mag(f0, 10, P)[2], P), P)
        assert 0.0 <= (c[0, 0] - f0) / f0 < 2.0 ** (1.0 / P.ppo) - 1.0
        assert c[0, 1] > so.VOICING_CUTOFF                              # resolved harmonics: voiced
    noise = np.random.default_rng(0).standard_normal(P.frame + 4 * P.hop).astype(np.float32) * 0.1
    c = so.shs_candidates(so.spec_scale(so.magnitudes(noise, P)[2][2], P), P)
    assert c[0, 1] < so.VOICING_CUTOFF                                  # white noise: below the cut-off
    z = so.shs_candidates(so.spec_scale(np.zeros(P.nbins), P), P)
    assert np.all(z == 0)                                               # silence: no candidates


# ---- cPitchSmootherViterbi ------------------------------------------------------------------------------
def _cands(T, f, v):
    c = np.zeros((T, 6, 3))
    c[:, 0, 0], c[:, 0, 1], c[:, 0, 2] = f, v, 1.0
    return c


def test_viterbi_removes_an_octave_outlier_and_keeps_voicing():
    T = 50
    c = _cands(T, 200.0, 0.85)
    c[20, 0, 0] = 400.0                                                 # best-scored candidate jumps an octave ...
    c[20, 1] = (200.0, 0.80, 0.9)                                       # ... the true pitch is the second candidate
    F, V = so.viterbi_smooth(c)
    assert np.allclose(F, 200.0) and abs(V[20] - 0.80) < 1e-12 and np.allclose(np.delete(V, 20), 0.85)


def test_viterbi_voicing_threshold_and_unclipped_output():
    T = 40
    c = _cands(T, 150.0, 0.5)                                           # below the 0.7 cut-off everywhere
    F, V = so.viterbi_smooth(c)
    assert np.all(F == 0.0) and np.allclose(V, 0.5)                     # unvoiced, voicing reported unclipped
    c = _cands(T, 150.0, 0.9)
    c[10:14, 0, 1] = 0.55                                               # a short dip: bridging it costs 4 x (wThr + local)
    F2, _ = so.viterbi_smooth(c)                                        # = 4 x 5.2 > two V/UV switches 2 x 10 + ...
    assert np.all(F2[:10] == 150.0) and np.all(F2[14:] == 150.0)
    e = so.viterbi_smooth(np.zeros((7, 6, 3)))
    assert np.all(e[0] == 0) and np.all(e[1] == 0)
    assert so.viterbi_smooth(np.zeros((0, 6, 3)))[0].shape == (0,)


def test_viterbi_fixed_lag_equals_full_backtrace_for_short_inputs():
    rng = np.random.default_rng(3)
    T = so.VIT_BUFLEN                                                   # every decision sees the last frame
    c = np.zeros((T, 6, 3))
    c[:, :3, 0] = rng.uniform(80, 400, (T, 3))
    c[:, :3, 1] = rng.uniform(0.4, 0.95, (T, 3))
    c[:, :3, 2] = 1.0
    F, _ = so.viterbi_smooth(c)
    c2 = np.concatenate([c, c[-1:]])                                    # one more frame changes the horizon of frame 0 only
    F2, _ = so.viterbi_smooth(c2)
    assert F.shape == (T,) and F2.shape == (T + 1,)
    assert set(np.unique(F)) <= set(np.unique(c[:, :, 0]))


def test_energy_gate():
    F, V = so.energy_gate(np.array([100.0, 120.0]), np.array([0.8, 0.9]), np.array([0.0009, 0.001]))
    assert list(F) == [0.0, 120.0] and list(V) == [0.0, 0.9]


# ---- cPitchJitter ----------------------------------------------------------------------------------------
def _pulse_train(periods, amps, n, width=6):
    x = np.zeros(n)
    pos = 40
    k = 0
    while pos + width < n:
        x[pos:pos + width] += amps[k % len(amps)] * np.hanning(width + 2)[1:-1]
        pos += periods[k % len(periods)]
        k += 1
    return x.astype(np.float32)


def test_jitter_zero_for_a_perfectly_periodic_signal():
    P = so.Params(16000)
    x = _pulse_train([80], [0.5], 16000)                                # 200 Hz, integer period
    nf = P.n_frames(len(x))
    J = so.jitter_shimmer(x, np.full(nf, 200.0), P)
    body = J[:, 2:-6]
    assert np.abs(body[0]).max() < 1e-9 and np.abs(body[1]).max() < 1e-9 and np.abs(body[2]).max() < 1e-9
    assert np.allclose(body[3], np.log(so.JIT_CC_MAX / (1 - so.JIT_CC_MAX)))    # cc* = 1 -> clipped
    assert np.all(J[:, -1] == J[:, -2])                                 # chain stopped at the clip end: values held


def test_jitter_and_shimmer_of_alternating_periods():
    P = so.Params(16000)
    x = _pulse_train([78, 82], [0.5, 0.4], 16000)
    nf = P.n_frames(len(x))
    J = so.jitter_shimmer(x, np.full(nf, 200.0), P)
    mid = J[:, 10:60]
    assert np.all(mid[0] > 0.02) and np.all(mid[0] < 0.08)              # |78 - 82| / 80 = 0.05 up to matching effects
    assert np.all(mid[2] > 0.1)                                         # 0.5 vs 0.4 peaks
    assert np.all(mid[1] > mid[0])                                      # DDP of an alternating sequence = 2 x local
    U = so.jitter_shimmer(x, np.zeros(nf), P)
    assert np.all(U == 0.0)                                             # unvoiced frames: zeros
    half = np.full(nf, 200.0)
    half[30:] = 0.0
    H = so.jitter_shimmer(x, half, P)
    assert np.all(H[:, 30:] == 0.0) and np.array_equal(H[:, :30], J[:, :30])    # causal: a run's past is unaffected


def test_lld_pitch_rows_on_a_voiced_tone():
    P = so.Params(16000)
    t = np.arange(16000) / 16000.0
    x = sum((1.0 / h) * np.sin(2 * np.pi * 250.0 * h * t) for h in range(1, 11))
    x = (0.3 * x / np.abs(x).max()).astype(np.float32)
    L = so.lld(x, P)
    F = L[so.I_F0]
    assert (F > 0).mean() > 0.95 and abs(np.median(F[F > 0]) - 250.0) / 250.0 < 0.025
    assert np.median(L[so.I_VOICE]) > 0.7 and np.median(L[so.I_JL]) < 1e-3 and np.median(L[so.I_HNR]) > 5.0
    q = so.lld(np.zeros(4000, dtype=np.float32), P)
    assert np.all(q[[14, 15, 18, 19, 20, 21]] == 0.0)                   # silence: gated, unvoiced, zeros
