"""HIP Praat-style MSHDS kernels vs the CPU oracle (parity unpinned: no Praat binary exists here)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mshds_oracle as mo
from robust_speech_analysis_framework_amd import synth

TOL = 1e-4


def _pack(clips):
    import torch
    lengths = [len(c) for c in clips]
    offs = np.concatenate([[0], np.cumsum(lengths)])[:-1]
    wav = torch.from_numpy(np.concatenate(clips)).cuda()
    return wav, [int(o) for o in offs], lengths


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.fixture(scope="module")
def eng(rsaf_lib):
    from robust_speech_analysis_framework_amd.mshds import MshdsEngine
    return MshdsEngine()


def test_intensity_contour_and_stats(eng):
    import torch
    clips = [synth.synth_clip(100, 2.0), synth.synth_clip(101, 1.3)]
    wav, offs, lens = _pack(clips)
    for floor in (60.0, 100.0):
        r = eng.intensity(wav, offs, lens, floor, 0.005, True)
        torch.cuda.synchronize()
        db = r["db"].cpu().numpy()
        st = r["stats"].cpu().numpy()
        for i, c in enumerate(clips):
            ref, t1, _ = mo.intensity(c, floor, 0.005, True)
            ci = r["ci"][i]
            assert ci["n_frames"] == len(ref) and abs(ci["t1"] - t1) < 1e-15       # integer-exact grid
            got = db[ci["frame_off"]:ci["frame_off"] + ci["n_frames"]]
            assert np.abs(got - ref).max() < 1e-8
            m, rr = mo.extract_intensity(c, floor, 0.005)
            assert abs(st[i, 0] - m) < 1e-8 and abs(st[i, 1] - rr) < 1e-8


@pytest.mark.parametrize("floor,ceil", [(50.0, 600.0), (100.0, 500.0), (60.0, 250.0)])
def test_pitch_ac_track_matches_oracle(eng, floor, ceil):
    import torch
    clips = [synth.synth_clip(110, 1.5), synth.synth_clip(111, 2.2)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    r = eng.pitch(wav, offs, lens, gp, time_step=0.005, floor=floor, ceiling=ceil)
    torch.cuda.synchronize()
    sel = r["sel_freq"].cpu().numpy()
    st = r["stats"].cpu().numpy()
    for i, c in enumerate(clips):
        p = mo.pitch_ac(c, 0.005, floor, pitch_ceiling=ceil)
        ci = r["ci"][i]
        assert ci["n_frames"] == p.n_frames
        got = sel[ci["frame_off"]:ci["frame_off"] + ci["n_frames"]]
        ref = p.frequency()
        assert np.array_equal(got > 0, ref > 0)                                  # voicing decisions identical
        assert _rel(got, ref) < 1e-7
        mean, sd = mo.extract_pitch(c, floor, ceil, 0.005, p)
        assert abs(st[i, 5] - mean) / mean < 1e-7 and abs(st[i, 6] - sd) / sd < 1e-6
        assert st[i, 4] == len(p.voiced_values())


@pytest.mark.parametrize("floor,ceil,dt", [(200.0, 800.0, 0.005), (400.0, 1600.0, 0.004), (30.0, 450.0, 0.02), (24.0, 300.0, 0.05)])
def test_pitch_ac_other_fft_lengths_match_oracle(eng, floor, ceil, dt):
    """The autocorrelation pass transforms with the smallest power of two >= 1.5 windows: the speaker ranges of the path use
    1 024 and 2 048 points; these floors run the 512-, 256- and 4 096-point instances of the FFT kernel (the last one
    with twiddles read from the table instead of registers, twice: 1 598- and 1 998-sample windows)."""
    import torch
    clips = [synth.synth_clip(120, 2.0), synth.synth_clip(121, 1.3)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    r = eng.pitch(wav, offs, lens, gp, time_step=dt, floor=floor, ceiling=ceil)
    torch.cuda.synchronize()
    sel = r["sel_freq"].cpu().numpy()
    for i, c in enumerate(clips):
        p = mo.pitch_ac(c, dt, floor, pitch_ceiling=ceil)
        ci = r["ci"][i]
        assert ci["n_frames"] == p.n_frames
        got = sel[ci["frame_off"]:ci["frame_off"] + ci["n_frames"]]
        ref = p.frequency()
        assert np.array_equal(got > 0, ref > 0)
        if (ref > 0).any():
            assert _rel(got, ref) < 1e-7


def test_pitch_dual_threshold_equals_two_separate_passes(eng):
    """:178 and :270 differ only in the voicing threshold: the dual launch must reproduce both standalone passes
    (candidate lists, selected track, statistics), including frames where a list overflows (max_candidates 4)."""
    import torch
    clips = [synth.synth_clip(112, 1.5), synth.synth_clip(113, 1.2)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    for kw in (dict(floor=75.0, ceiling=500.0), dict(floor=60.0, ceiling=250.0, max_candidates=4)):
        a = eng.pitch(wav, offs, lens, gp, time_step=0.005, voicing_threshold=0.45, **kw)
        b = eng.pitch(wav, offs, lens, gp, time_step=0.005, voicing_threshold=0.3, **kw)
        d = eng.pitch(wav, offs, lens, gp, time_step=0.005, voicing_threshold=0.45, voicing_threshold2=0.3, **kw)
        torch.cuda.synchronize()
        for single, dual in ((a, d), (b, d["second"])):
            fs, fd = single["frame_out"].cpu().numpy(), dual["frame_out"].cpu().numpy()
            assert np.array_equal(fs == 0, fd == 0)
            assert np.abs(fs - fd).max() <= 1e-9 * np.abs(fs).max()               # group size changes the sum order only
            ss, sd = single["sel_freq"].cpu().numpy(), dual["sel_freq"].cpu().numpy()
            assert np.array_equal(ss > 0, sd > 0) and np.abs(ss - sd).max() <= 1e-9 * ss.max()
            assert np.allclose(single["stats"].cpu().numpy(), dual["stats"].cpu().numpy(), rtol=1e-9, equal_nan=True)
    for i, c in enumerate(clips):                                                     # and the oracle at 0.3
        p = mo.pitch_ac(c, 0.005, 75.0, voicing_threshold=0.3, pitch_ceiling=500.0)
        d = eng.pitch(wav, offs, lens, gp, time_step=0.005, floor=75.0, ceiling=500.0, voicing_threshold2=0.3)["second"]
        ci = d["ci"][i]
        got = d["sel_freq"].cpu().numpy()[ci["frame_off"]:ci["frame_off"] + ci["n_frames"]]
        assert np.array_equal(got > 0, p.frequency() > 0) and _rel(got, p.frequency()) < 1e-7


def test_pitch_in_clip_groups_equals_one_launch(eng):
    """The correlation rows live in a caller workspace; when it holds fewer clips than the batch, the two pitch kernels
    run group by group.  Results must not depend on the grouping (ragged batch, groups of one clip)."""
    import torch
    clips = [synth.synth_clip(114, 1.0), synth.synth_clip(115, 0.6), synth.synth_clip(116, 1.3)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    kw = dict(time_step=0.005, floor=75.0, ceiling=500.0, voicing_threshold2=0.3)
    a = eng.pitch(wav, offs, lens, gp, **kw)
    cap = eng.pitch_ws_cap_bytes
    try:
        eng.pitch_ws_cap_bytes = 1.0                                  # -> exactly one clip per group
        b = eng.pitch(wav, offs, lens, gp, **kw)
    finally:
        eng.pitch_ws_cap_bytes = cap
    torch.cuda.synchronize()
    for x, y in ((a, b), (a["second"], b["second"])):
        assert torch.equal(x["frame_out"], y["frame_out"]) and torch.equal(x["sel_freq"], y["sel_freq"])
        assert torch.equal(torch.nan_to_num(x["stats"]), torch.nan_to_num(y["stats"]))


def test_pitch_ac_few_candidates_replacement_rule(eng):
    """The 4-candidate pass of _speechrate (:104): more maxima than slots -> weakest is replaced."""
    import torch
    clips = [synth.synth_clip(112, 2.0)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    kw = dict(time_step=0.02, floor=30.0, ceiling=450.0, max_candidates=4, silence_threshold=0.03,
              voicing_threshold=0.25, octave_cost=0.01, octave_jump_cost=0.35, voiced_unvoiced_cost=0.25)
    r = eng.pitch(wav, offs, lens, gp, **kw)
    torch.cuda.synchronize()
    p = mo.pitch_ac(clips[0], 0.02, 30.0, 4, False, 0.03, 0.25, 0.01, 0.35, 0.25, 450.0)
    got = r["sel_freq"].cpu().numpy()[:p.n_frames]
    assert r["ci"][0]["n_frames"] == p.n_frames
    assert np.array_equal(got > 0, p.frequency() > 0) and _rel(got, p.frequency()) < 1e-7


def test_harmonicity_cc_mean(eng):
    import torch
    clips = [synth.synth_clip(120, 1.2), synth.synth_clip(121, 1.6)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    for floor in (100.0, 60.0):
        cc = eng.pitch(wav, offs, lens, gp, time_step=0.005, floor=floor, ceiling=8000.0, max_candidates=15,
                       silence_threshold=0.1, voicing_threshold=0.0, octave_cost=0.0, octave_jump_cost=0.0,
                       voiced_unvoiced_cost=0.0, periods=4.5, is_cc=True, refine_depth=700)
        h = eng.hnr_mean(cc)
        torch.cuda.synchronize()
        for i, c in enumerate(clips):
            ref = mo.extract_harmonicity(c, floor, None, 0.005)
            assert abs(h[i].item() - ref) < 1e-6 * max(1.0, abs(ref)), (floor, i, h[i].item(), ref)


def test_spectral_moments_gated_by_pitch(eng):
    import torch
    clips = [synth.synth_clip(130, 1.5), synth.synth_clip(131, 2.0)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    p = eng.pitch(wav, offs, lens, gp, time_step=0.005, floor=100.0, ceiling=500.0)
    sm = eng.spectral_moments(wav, offs, lens, p, 0.025, 0.005)
    torch.cuda.synchronize()
    assert abs(sm["fstep"] - 15.625) < 1e-12
    got = sm["stats"].cpu().numpy()
    for i, c in enumerate(clips):
        ref = np.array(mo.extract_spectral_moments(c, 100, 500, 0.025, 0.005))
        assert np.abs(got[i] - ref).max() / np.abs(ref).max() < 1e-7, (got[i], ref)
        pw, t1, ts, fs = mo.spectrogram_power(c, 0.025, 5000.0, 0.005, 20.0)
        assert sm["ci"][i]["n_frames"] == pw.shape[0] and abs(sm["ci"][i]["t1"] - t1) < 1e-15


def test_speechrate_matches_oracle(eng):
    """_speechrate (:11-125): silence TextGrid, syllable nuclei, voiced & sounding count."""
    import torch
    clips = [synth.synth_clip(160, 6.0), synth.synth_clip(161, 3.5), np.zeros(8000, np.float32),
             synth.synth_clip(162, 0.1)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    got = eng.speechrate(wav, offs, lens, gp)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for i, c in enumerate(clips):
        ref = np.array(mo.speechrate(c), dtype=np.float64)
        assert np.array_equal(np.isnan(got[i]), np.isnan(ref)), (i, got[i], ref)
        ok = ~np.isnan(ref)
        assert np.allclose(got[i][ok], ref[ok], rtol=1e-9, atol=1e-12), (i, got[i], ref)
    assert got[0][0] > 1.0 and 0.3 < got[0][2] <= 1.0            # syllables/s and phonation ratio are sensible


def test_formants_resampler_pulses_and_statistics(eng):
    """_measureFormants (:303-338): 10 kHz resampling, Burg + roots per frame, cc pulses, per-pulse stats."""
    import torch
    clips = [synth.synth_clip(170, 1.5), synth.synth_clip(171, 1.0003125)]       # 2nd: odd resampled length
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    got = eng.formants(wav, offs, lens, gp, 100.0, 500.0, 0.005)
    torch.cuda.synchronize()
    L = eng._last_formants
    y10 = L["y10"].cpu().numpy()
    fr = L["frames"].cpu().numpy().reshape(-1, 10)
    pul = L["pulses"].cpu().numpy().reshape(len(clips), L["max_pulses"])
    npul = L["n_pulses"].cpu().numpy()
    for i, c in enumerate(clips):
        yr, x1o, dxo = mo.resample_10k(c)
        ri = L["ri"][i]
        assert ri["n_out"] == len(yr) and abs(ri["x1o"] - x1o) < 1e-18
        assert np.abs(y10[ri["out_off"]:ri["out_off"] + ri["n_out"]] - yr).max() < 1e-7   # oracle positions carry ~1e-11 rel rounding
        F, B, t1, dt = mo.formant_burg(c)
        ci = L["ci"][i]
        assert ci["n_frames"] == F.shape[0] and abs(ci["t1"] - t1) < 1e-15
        g = fr[ci["frame_off"]:ci["frame_off"] + ci["n_frames"]]
        assert np.array_equal(np.isnan(g[:, :5]), np.isnan(F))
        ok = ~np.isnan(F)
        # polynomial roots amplify the last-bit differences of the Burg sums: compare at 2e-6 of 5 kHz
        assert np.abs(g[:, :5][ok] - F[ok]).max() < 1e-2 and np.abs(g[:, 5:][ok] - B[ok]).max() < 1e-2
        p = mo.pitch_cc(c, 0.005, 100.0, 1.0, 15, 0.03, 0.45, 0.01, 0.35, 0.14, 500.0)
        pts = mo.point_process_cc(c.astype(np.float64), p)
        assert npul[i] == len(pts)                                               # integer-exact pulse count
        assert np.abs(pul[i, :npul[i]] - pts).max() < 1e-9                       # ascending time, like a PointProcess
        ref = np.array(mo.measure_formants(c, 100, 500))
        assert np.abs(got[i].cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-6, (got[i], ref)


def test_ltas_slope_and_tilt(eng):
    """_extract_Slope_Tilt (:227-251): AC pitch with the automatic time step, cc pulses, one-period spectra
    binned into 100 Hz bands, "Get slope" and the robust (Theil) tilt."""
    import torch
    clips = [synth.synth_clip(180, 2.0), synth.synth_clip(181, 1.3), np.zeros(4000, np.float32),
             synth.synth_clip(182, 0.05)]                                         # silence / too short: NaN, NaN
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    for floor, ceil in ((60.0, 250.0), (100.0, 500.0)):
        got = eng.slope_tilt(wav, offs, lens, gp, floor, ceil).cpu().numpy()
        torch.cuda.synchronize()
        L = eng._last_ltas
        pul = L["pulses"].cpu().numpy().reshape(len(clips), L["max_pulses"])
        npul = L["n_pulses"].cpu().numpy()
        for i, c in enumerate(clips):
            p = mo.pitch_ac(c, 0.0, floor, pitch_ceiling=ceil)
            pts = mo.point_process_cc(c.astype(np.float64), p)
            assert npul[i] == len(pts)
            if len(pts):
                assert np.abs(pul[i, :npul[i]] - pts).max() < 1e-9
            ref = np.array(mo.extract_slope_tilt(c, floor, ceil))
            assert np.array_equal(np.isnan(got[i]), np.isnan(ref)), (i, got[i], ref)
            if not np.isnan(ref).any():
                assert abs(got[i, 0] - ref[0]) <= 1e-7 * abs(ref[0]) + 1e-9, (got[i], ref)
                assert abs(got[i, 1] - ref[1]) <= 1e-6 * abs(ref[1]) + 1e-12, (got[i], ref)
    assert not np.isnan(got[0]).any() and np.isnan(got[2]).all()


def test_cpp_voiced_intervals_cepstrogram_and_cpps(eng):
    """_extract_CPP (:253-301): vuv intervals (6-decimal times), 10 kHz resampling, power cepstrum per frame,
    smoothed CPP per frame, mean of the per-interval CPPS above 4 dB."""
    import torch
    # third clip: odd sample count and voiced to the end -> the last interval ends at n/16000 s, which lies within an
    # ulp of a 6-decimal tie (the "Down to Table" rounding must follow printf on the exact binary value)
    tie = synth.synth_clip(5004, 1.6078125)
    assert len(tie) % 2 == 1
    clips = [synth.synth_clip(190, 1.6), synth.synth_clip(191, 1.1), tie, np.zeros(4000, np.float32)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    got = eng.cpp(wav, offs, lens, gp, 100.0, 500.0).cpu().numpy()
    torch.cuda.synchronize()
    L = eng._last_cpp
    sd = L["seg_doubles"]
    segs = L["segs"].cpu().numpy().reshape(len(clips), L["max_seg"], sd)
    hdr = L["hdr"].cpu().numpy().reshape(len(clips), 4)
    res = L["res"].cpu().numpy().reshape(len(clips), L["cap_res"])
    ceps = L["ceps"].cpu().numpy().reshape(len(clips), L["cap_frames"], 513)
    cppf = L["cpp_frames"].cpu().numpy().reshape(len(clips), L["cap_frames"])
    for i, c in enumerate(clips):
        x = c.astype(np.float64)
        p = mo.pitch_ac(x, 0.005, 100.0, voicing_threshold=0.3, pitch_ceiling=500.0)
        pul = mo.point_process_cc(x, p)
        iv = [(float(f"{a:.6f}"), float(f"{b:.6f}")) for a, b in mo.vuv_intervals(pul, 0.0, len(x) * mo.DX)]
        iv = [(a, b) for a, b in iv if a < b]
        assert hdr[i, 0] == len(iv) and hdr[i, 1] == 0                               # interval count is exact
        ref_vals = []
        for k, (tmin, tmax) in enumerate(iv):
            ix1 = int(np.ceil((tmin - 0.5 * mo.DX) / mo.DX))
            ix2 = int(np.floor((tmax - 0.5 * mo.DX) / mo.DX))
            seg = np.zeros(ix2 - ix1 + 1)
            a, b = max(ix1, 0), min(ix2, len(x) - 1)
            seg[a - ix1:b - ix1 + 1] = x[a:b + 1]
            x1_seg = 0.5 * mo.DX + ix1 * mo.DX - tmin
            S = segs[i, k]
            assert int(S[0]) == ix1 and int(S[1]) == len(seg)                         # integers exact
            y, x1o, _ = mo.resample_part(seg, x1_seg, tmax - tmin, mo.CPP_FS, mo.CPP_DEPTH)
            assert int(S[2]) == len(y) and abs(S[7] - x1o) < 1e-15
            r0 = int(S[3])
            assert np.abs(res[i, r0:r0 + len(y)] - y).max() < 1e-9 * max(1.0, np.abs(y).max())
            z = mo.power_cepstrogram(seg, x1_seg, tmax - tmin)
            f0, nf = int(S[4]), int(S[5])
            assert nf == z.shape[1] and int(S[11]) // 2 + 1 == z.shape[0]
            g = ceps[i, f0:f0 + nf, :z.shape[0]].T
            assert np.abs(g - z).max() <= 1e-7 * np.abs(z).max()
            v = mo.cpps(z)
            assert abs(cppf[i, f0:f0 + nf].mean() - v) <= 1e-6 * abs(v)
            if v > 4:
                ref_vals.append(v)
        ref = np.mean(ref_vals) if ref_vals else np.nan
        assert np.isnan(got[i]) == np.isnan(ref)
        if not np.isnan(ref):
            assert abs(got[i] - ref) <= 1e-6 * abs(ref), (got[i], ref)
            assert abs(ref - mo.extract_cpp(c, 100.0, 500.0)) < 1e-12
    assert not np.isnan(got[0]) and np.isnan(got[3])


def test_extract_packed_matches_oracle_and_uses_both_speaker_ranges(eng):
    import torch
    ids = [140, 141, 142, 143, 144, 145]
    clips = [synth.synth_clip(k, 1.5) for k in ids] + [np.zeros(300, np.float32)]
    wav, offs, lens = _pack(clips)
    out, ranges = eng.extract_packed(wav, offs, lens)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    seen = set()
    for i, c in enumerate(clips):
        ref, rng = mo.extract(c)
        assert tuple(rng) == tuple(ranges[i])                                   # integer outputs exact
        seen.add(tuple(rng))
        assert np.array_equal(np.isnan(got[i]), np.isnan(ref)), (i, got[i], ref)
        ok = ~np.isnan(ref)
        if ok.any():
            assert np.abs(got[i][ok] - ref[ok]).max() <= TOL * np.abs(ref[ok]).max()
            assert (np.abs(got[i][ok] - ref[ok]) <= TOL * np.maximum(np.abs(ref[ok]), 1e-3)).all(), (i, got[i][ok], ref[ok])
    assert {(60, 250), (100, 500)} <= seen and (75, 500) in seen


def test_dropin_dataframe_contract(eng, tmp_path):
    import pandas as pd
    from robust_speech_analysis_framework_amd.mshds import FEATURE_NAMES, extract_mshds_features
    paths = synth.write_synth_corpus(str(tmp_path), 2, 1.2, first=150)
    bad = tmp_path / "broken.wav"
    bad.write_bytes(b"junk")
    df = pd.DataFrame({"filepath": [paths[0], str(bad), paths[1]]})
    out = extract_mshds_features(df, verbose=False)
    assert list(out.columns) == ["filename"] + FEATURE_NAMES and len(out) == 3
    assert list(out["filename"]) == ["synth_00150.wav", "broken.wav", "synth_00151.wav"]
    assert out.iloc[1, 1:].isna().all()                                         # failed file -> NaN row (:450-457)
    ref, _ = mo.extract(synth.synth_clip(150, 1.2))
    row = out.iloc[0, 1:].to_numpy(dtype=np.float64)
    ok = ~np.isnan(ref)
    assert np.array_equal(np.isnan(row), ~ok)
    assert (np.abs(row[ok] - ref[ok]) <= TOL * np.maximum(np.abs(ref[ok]), 1e-3)).all()


def _oracle_extract(args):
    k, seconds = args
    from oracle import mshds_oracle
    from robust_speech_analysis_framework_amd import synth as sy
    r, rng = mshds_oracle.extract(sy.synth_clip(k, seconds))
    return r, tuple(rng)


def _oracle_pool(jobs):
    """The numpy restatement takes ~1.4 s per audio-second: run the clips of a config in worker processes
    (spawned: the test process has initialised the GPU and must not fork)."""
    import concurrent.futures as cf
    import multiprocessing as mp
    with cf.ProcessPoolExecutor(max_workers=min(8, len(jobs)), mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(_oracle_extract, jobs))


def _check_rows(got, refs):
    for i, (ref, _) in enumerate(refs):
        assert np.array_equal(np.isnan(got[i]), np.isnan(ref)), (i, got[i], ref)
        ok = ~np.isnan(ref)
        assert ok.sum() >= 20                                                   # a voiced clip defines the features
        assert (np.abs(got[i][ok] - ref[ok]) <= TOL * np.maximum(np.abs(ref[ok]), 1e-3)).all(), (i, got[i][ok], ref[ok])


def test_config_c1_ten_5s_wav_files_through_the_dropin(eng, tmp_path):
    """BASELINE config C1: MSHDS 25-feature extract on 10 synthetic 16 kHz 5 s mono WAVs through the reference's entry
    point (src.mshds_extractor.extract_mshds_features); all 25 columns + NaN pattern against the CPU restatement."""
    import pandas as pd
    from src.mshds_extractor import extract_mshds_features
    from robust_speech_analysis_framework_amd.mshds import FEATURE_NAMES
    paths = synth.write_synth_corpus(str(tmp_path), 10, 5.0, first=20260000)
    out = extract_mshds_features(pd.DataFrame({"filepath": paths}), verbose=False)
    assert list(out.columns) == ["filename"] + FEATURE_NAMES and len(out) == 10
    assert list(out["filename"]) == [f"synth_{20260000 + k:05d}.wav" for k in range(10)]
    refs = _oracle_pool([(20260000 + k, 5.0) for k in range(10)])
    _check_rows(out.iloc[:, 1:].to_numpy(dtype=np.float64), refs)


def test_config_c2_mshds_30s_clips_match_oracle(eng):
    """BASELINE config C2 clip length: full 30 s clips (5 990 pitch frames, ~25 voiced stretches each) through
    extract_packed in one batch, against the CPU restatement, including the speaker-range decision."""
    import torch
    ids = [20260100, 20260101, 20260102]
    clips = [synth.synth_clip(k, 30.0) for k in ids]
    wav, offs, lens = _pack(clips)
    out, ranges = eng.extract_packed(wav, offs, lens)
    torch.cuda.synchronize()
    refs = _oracle_pool([(k, 30.0) for k in ids])
    for i, (_, rng) in enumerate(refs):
        assert tuple(ranges[i]) == rng
    _check_rows(out.cpu().numpy(), refs)


# ---- files that the reference resamples first (src/mshds_extractor.py:418-419): the 16 kHz sound then carries Praat's
# centred time axis (x1 != dx / 2, xmax = the ORIGINAL duration) through every analysis -------------------------------
def _oracle_extract_resampled(args):
    k, seconds, fs = args
    from oracle import mshds_oracle, resample_oracle
    from robust_speech_analysis_framework_amd import synth as sy
    y, x1, xmax = resample_oracle.resample_praat_sound(sy.synth_clip(k, seconds, fs=fs), float(fs), 16000.0, 50)
    r, rng = mshds_oracle.extract(y, x1, xmax)
    return r, tuple(rng), y, x1, xmax


def _oracle_pool_resampled(jobs):
    import concurrent.futures as cf
    import multiprocessing as mp
    with cf.ProcessPoolExecutor(max_workers=min(8, len(jobs)), mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(_oracle_extract_resampled, jobs))


RESAMPLED_JOBS = [(20260300, 5.0, 44100), (20260301, 5.0, 8000), (20260302, 3.70001, 44100), (20260303, 2.0, 22050),
                  (20260304, 2.5, 48000), (20260305, 1.3001, 11025)]


@pytest.fixture(scope="module")
def resampled_refs():
    return _oracle_pool_resampled(RESAMPLED_JOBS)


def test_extract_packed_with_praats_resampled_time_axis(eng, resampled_refs):
    """The SAME float32 samples through both paths (the oracle's resampled sounds), with the time axis Praat gives them:
    all 25 columns, the NaN pattern and the speaker range; and the axis matters: x1 differs from dx / 2 for these files."""
    import torch
    clips = [r[2] for r in resampled_refs]
    x1s, xmaxs = [r[3] for r in resampled_refs], [r[4] for r in resampled_refs]
    assert max(abs(x1 - 0.5 / 16000.0) for x1 in x1s) > 1e-6                    # off the file grid by up to a quarter sample
    assert max(abs(xm - len(c) / 16000.0) for xm, c in zip(xmaxs, clips)) > 1e-6  # the domain is not nx dx
    wav, offs, lens = _pack(clips)
    out, ranges = eng.extract_packed(wav, offs, lens, x1=x1s, xmax=xmaxs)
    torch.cuda.synchronize()
    for i, r in enumerate(resampled_refs):
        assert tuple(ranges[i]) == r[1]
    _check_rows(out.cpu().numpy(), [(r[0], r[1]) for r in resampled_refs])


def test_dropin_on_44k1_and_8k_files_matches_resample_then_extract(eng, tmp_path, resampled_refs):
    """src.mshds_extractor on WAV files at 44.1 kHz (the Androids corpus rate, Androids.conf:70), 8 kHz (Sound_upsample),
    22.05 / 48 / 11.025 kHz: device decode -> device Sound_resample -> analyses on Praat's centred grid, against
    resample_oracle -> mshds_oracle(x1, xmax).  All 25 columns + the NaN pattern."""
    import pandas as pd
    from src.mshds_extractor import extract_mshds_features
    paths = []
    for k, seconds, fs in RESAMPLED_JOBS:
        p = str(tmp_path / f"clip_{k}_{fs}.wav")
        synth.write_wav(p, synth.synth_clip_int16(k, seconds, fs), fs=fs)
        paths.append(p)
    out = extract_mshds_features(pd.DataFrame({"filepath": paths}), verbose=False)
    assert len(out) == len(paths)
    _check_rows(out.iloc[:, 1:].to_numpy(dtype=np.float64), [(r[0], r[1]) for r in resampled_refs])


def test_time_axis_shifts_times_but_not_windows(eng):
    """A sound whose axis is shifted by a constant (x1 and xmax moved together) has the same sample windows in every
    analysis: time-free features are unchanged, and the default axis equals the explicit file axis bit for bit."""
    import torch
    c = synth.synth_clip(146, 1.5)
    wav, offs, lens = _pack([c, c, c])
    d = 0.2 / 16000.0
    out, _ = eng.extract_packed(wav, offs, lens, x1=[0.5 / 16000.0, 0.5 / 16000.0 + d, 0.5 / 16000.0 - d],
                                xmax=[len(c) / 16000.0, len(c) / 16000.0 + d, len(c) / 16000.0 - d])
    ref, _ = eng.extract_packed(wav[:len(c)].contiguous(), [0], [len(c)])
    torch.cuda.synchronize()
    g, r = out.cpu().numpy(), ref.cpu().numpy()[0]
    assert np.array_equal(g[0], r, equal_nan=True)
    # mean F0 / SD, HNR, spectral moments take the LOW sample index of half-sample frame times (robust); intensity and the
    # pulse walker take the NEAREST sample of exact ties, where the last bit of (t - x1) / dx decides as it does in Praat
    cols = [5, 6, 9, 21, 22, 23, 24]
    for i in (1, 2):
        assert np.array_equal(np.isnan(g[i]), np.isnan(r))
        assert (np.abs(g[i][cols] - r[cols]) <= 1e-9 * np.maximum(np.abs(r[cols]), 1e-3)).all(), (g[i], r)


def test_one_wave_pitch_kernels_agree_with_the_workgroup_kernels_on_30s_clips(eng, monkeypatch):
    """The correlation kernels exist twice: one wavefront per frame with the transform in registers (csrc/wave_fft.h, the
    product path) and the workgroup-wide Stockham kernels of round 2 (RSAF_PITCH_FFT=wg).  Two independent implementations
    of the same arithmetic must pick the same path through every frame of full-length clips (6 000 frames each) for every
    parameter set the extractor uses - a size-independent check at BASELINE's clip length, where the oracle is too slow."""
    import torch
    from robust_speech_analysis_framework_amd import _lib
    clips = [synth.synth_clip(900 + k, 30.0) for k in range(3)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    hnr = dict(max_candidates=15, silence_threshold=0.1, voicing_threshold=0.0, octave_cost=0.0, octave_jump_cost=0.0,
               voiced_unvoiced_cost=0.0, periods=4.5, is_cc=True, refine_depth=700)
    cfgs = [dict(time_step=0.005, floor=50.0, ceiling=600.0),                                  # wide AC, 1 024 complex points
            dict(time_step=0.005, floor=100.0, ceiling=500.0, voicing_threshold2=0.3),         # 512 points, both thresholds
            dict(time_step=0.02, floor=30.0, ceiling=450.0, max_candidates=4, voicing_threshold=0.25,
                 voiced_unvoiced_cost=0.25),                                                    # speech rate: 2 048 points
            dict(time_step=0.005, floor=60.0, ceiling=8000.0, **hnr),                           # CC, 2 048 -> 1 024 points
            dict(time_step=0.005, floor=100.0, ceiling=8000.0, **hnr),                          # CC, 1 024 -> 512 points
            dict(time_step=0.005, floor=100.0, ceiling=500.0, periods=1.0, is_cc=True, refine_depth=70)]   # padded up to 1 024
    for kw in cfgs:
        monkeypatch.delenv("RSAF_PITCH_FFT", raising=False)
        a = eng.pitch(wav, offs, lens, gp, **kw)
        torch.cuda.synchronize()
        monkeypatch.setenv("RSAF_PITCH_FFT", "wg")
        try:
            b = eng.pitch(wav, offs, lens, gp, **kw)
        except _lib.RsafError as e:
            if "RSAF_BUILD_TEST_KERNELS" in str(e):
                monkeypatch.delenv("RSAF_PITCH_FFT", raising=False)
                pytest.skip("the superseded workgroup-FFT kernels are only in a test build (RSAF_BUILD_TEST_KERNELS=1); the one-wave "
                            "kernels are checked frame by frame against the oracle in test_known_answers_gpu.py")
            raise
        torch.cuda.synchronize()
        fa, fb = a["sel_freq"].cpu().numpy(), b["sel_freq"].cpu().numpy()
        sa, sb = a["sel_strength"].cpu().numpy(), b["sel_strength"].cpu().numpy()
        assert fa.shape == fb.shape and fa.size >= 3 * 1400
        # The two forms differ by rounding (~1e-15 in the correlation), which can only matter where the path finder meets a
        # numerical tie: allow one frame in a thousand to take another candidate, nothing else
        differs = ((fa > 0) != (fb > 0)) | (np.abs(fa - fb) > 1e-7 * np.maximum(np.abs(fb), 1.0)) | (np.abs(sa - sb) > 1e-9)
        print(f"pitch A/B floor {kw['floor']:g} cc {kw.get('is_cc', False)}: {int(differs.sum())} of {fa.size} frames differ")
        assert differs.mean() <= 1e-3, (kw, int(differs.sum()), fa.size)
    monkeypatch.delenv("RSAF_PITCH_FFT", raising=False)


def test_refinement_kernels_agree_with_the_in_kernel_refinement(eng, monkeypatch):
    """Candidate refinement exists twice: inside the candidate kernel (RSAF_PITCH_INKERNEL=1: one wave per frame, the form
    before round 4; the harmonicity pass at a 100 Hz floor evaluates Praat's clipped sinc sums directly there) and as the
    product's pipeline candidate kernel -> [per-cell coefficient GEMM] -> Brent kernel (one candidate per lane).  Every
    candidate of every frame must come out the same: identical bits where both forms search the same polynomial (the AC
    passes, both thresholds of the dual pass, the depth-70 pulse pass), within the Chebyshev fit (1e-10) where the
    in-kernel form is the direct sum or accumulates the taps in another order (harmonicity passes)."""
    import torch
    clips = [synth.synth_clip(930 + k, 8.0) for k in range(3)] + [synth.synth_clip(77, 0.9), np.zeros(4000, np.float32)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    hnr = dict(max_candidates=15, silence_threshold=0.1, voicing_threshold=0.0, octave_cost=0.0, octave_jump_cost=0.0,
               voiced_unvoiced_cost=0.0, periods=4.5, is_cc=True, refine_depth=700)
    cfgs = [(dict(time_step=0.005, floor=50.0, ceiling=600.0), True),
            (dict(time_step=0.005, floor=100.0, ceiling=500.0, voicing_threshold2=0.3), True),
            (dict(time_step=0.005, floor=60.0, ceiling=250.0, voicing_threshold2=0.3), True),
            (dict(time_step=0.005, floor=100.0, ceiling=500.0, voicing_threshold2=0.6), True),     # second threshold ABOVE the first
            (dict(time_step=0.02, floor=30.0, ceiling=450.0, max_candidates=4, voicing_threshold=0.25, voiced_unvoiced_cost=0.25), True),
            (dict(time_step=0.005, floor=100.0, ceiling=500.0, periods=1.0, is_cc=True, refine_depth=70), True),
            (dict(time_step=0.005, floor=60.0, ceiling=8000.0, **hnr), False),
            (dict(time_step=0.005, floor=75.0, ceiling=8000.0, **hnr), False),
            (dict(time_step=0.005, floor=100.0, ceiling=8000.0, **hnr), False)]
    fo = eng.fo_doubles
    for kw, exact in cfgs:
        monkeypatch.delenv("RSAF_PITCH_INKERNEL", raising=False)
        a = eng.pitch(wav, offs, lens, gp, **kw)
        torch.cuda.synchronize()
        monkeypatch.setenv("RSAF_PITCH_INKERNEL", "1")
        b = eng.pitch(wav, offs, lens, gp, **kw)
        torch.cuda.synchronize()
        monkeypatch.delenv("RSAF_PITCH_INKERNEL", raising=False)
        pairs = [(a, b)] + ([(a["second"], b["second"])] if "voicing_threshold2" in kw else [])
        for ra, rb in pairs:
            n = ra["total_frames"]
            A = ra["frame_out"].cpu().numpy()[:n * fo].reshape(n, fo)
            B = rb["frame_out"].cpu().numpy()[:n * fo].reshape(n, fo)
            assert n > 1000 and np.array_equal(A[:, :2], B[:, :2])                   # intensity, candidate count
            assert np.array_equal(A[:, 2:18] > 0, B[:, 2:18] > 0)                    # the same lists
            if exact:
                assert np.array_equal(A, B), kw
            else:
                assert np.abs(A[:, 18:] - B[:, 18:]).max() <= 1e-10, kw              # strengths
                on = A[:, 2:18] > 0                                                  # positions: within Brent's own stopping
                lag_a, lag_b = 16000.0 / A[:, 2:18][on], 16000.0 / B[:, 2:18][on]    # tolerance (2 sqrt(eps) x, x <= 2 500 samples)
                assert np.abs(lag_a - lag_b).max() <= 1e-4, kw
            sa, sb = ra["stats"].cpu().numpy(), rb["stats"].cpu().numpy()
            assert np.allclose(sa, sb, rtol=1e-9, atol=0, equal_nan=True)


def test_one_wave_cpps_kernels_agree_with_the_workgroup_kernels(eng, monkeypatch):
    """CPPS frames of full-window intervals run one wavefront per frame (register transforms, medians by rank selection).
    RSAF_CPP_WAVE=0 sends every frame through the workgroup kernels (bit-reversal FFT in LDS, bitonic sorts), =c only the
    smoothed-CPP frames.  The cepstra of the two forms differ in rounding (different butterfly order); the frame kernel is
    the same arithmetic in the same order on either form (fp contraction off), so on ONE cepstrogram it must return
    identical bits - medians by selection = medians by sorting."""
    import torch
    clips = [synth.synth_clip(910 + k, 12.0) for k in range(3)] + [synth.synth_clip(190, 1.6)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    out = {}
    for mode in ("1", "c", "0"):
        monkeypatch.setenv("RSAF_CPP_WAVE", mode)
        got = eng.cpp(wav, offs, lens, gp, 100.0, 500.0).cpu().numpy()
        torch.cuda.synchronize()
        L = eng._last_cpp
        hdr = L["hdr"].cpu().numpy().reshape(len(clips), 4)
        segs = L["segs"].cpu().numpy().reshape(len(clips), L["max_seg"], L["seg_doubles"])
        cppf = L["cpp_frames"].cpu().numpy().reshape(len(clips), L["cap_frames"])
        ceps = L["ceps"].cpu().numpy().reshape(len(clips), L["cap_frames"], 513)
        rows = []
        for i in range(len(clips)):
            c = ceps[i, :hdr[i, 2]].copy()
            for k in range(hdr[i, 0]):                            # a short interval's frames hold nfft / 2 + 1 bins: the rest of
                f0, nf, nfft = int(segs[i, k, 4]), int(segs[i, k, 5]), int(segs[i, k, 11])   # the row is never written
                c[f0:f0 + nf, nfft // 2 + 1:] = 0.0
            rows.append(c)
        out[mode] = (got, [cppf[i, :hdr[i, 2]].copy() for i in range(len(clips))], rows)
    monkeypatch.delenv("RSAF_CPP_WAVE", raising=False)
    assert sum(len(f) for f in out["1"][1]) > 5000                                             # thousands of frames compared
    for i in range(len(clips)):
        assert np.array_equal(out["1"][2][i], out["c"][2][i])                                  # same cepstrum kernel: same bits
        assert np.array_equal(out["1"][1][i], out["c"][1][i], equal_nan=True)                  # frame kernel: identical bits
        c1, c0 = out["1"][2][i], out["0"][2][i]
        assert c1.shape == c0.shape and np.abs(c1 - c0).max() <= 1e-9 * max(np.abs(c0).max(), 1e-300)
    assert np.array_equal(out["1"][0], out["c"][0], equal_nan=True)
    assert np.array_equal(np.isnan(out["1"][0]), np.isnan(out["0"][0]))
    ok = ~np.isnan(out["0"][0])
    assert np.abs(out["1"][0][ok] - out["0"][0][ok]).max() <= 1e-9 * np.abs(out["0"][0][ok]).max()


def test_one_wave_spectral_moments_agree_with_the_workgroup_kernel(eng, monkeypatch):
    """Spectral moments per frame by one wavefront (register transform, csrc/wave_fft.h) against the 256-thread radix-2
    kernel (RSAF_SPM_WAVE=0) on 30 s clips: the gate pattern identical, every moment of every frame within 1e-9."""
    import torch
    clips = [synth.synth_clip(920 + k, 30.0) for k in range(2)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    p = eng.pitch(wav, offs, lens, gp, time_step=0.005, floor=100.0, ceiling=500.0)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RSAF_SPM_WAVE", mode)
        sm = eng.spectral_moments(wav, offs, lens, p, 0.025, 0.005)
        torch.cuda.synchronize()
        res[mode] = (sm["moments"].cpu().numpy().reshape(-1, 5).copy(), sm["stats"].cpu().numpy().copy())
    monkeypatch.delenv("RSAF_SPM_WAVE", raising=False)
    m1, m0 = res["1"][0], res["0"][0]
    assert m1.shape == m0.shape and m1.shape[0] > 10000
    assert np.array_equal(m1[:, 0], m0[:, 0]) and m1[:, 0].sum() > 1000                       # gate: same frames analysed
    on = m0[:, 0] == 1.0
    assert np.abs(m1[on, 1:] - m0[on, 1:]).max() <= 1e-9 * np.abs(m0[on, 1:]).max()
    assert np.abs(res["1"][1] - res["0"][1]).max() <= 1e-9 * np.abs(res["0"][1]).max()


def test_cpps_polyphase_resampling_agrees_with_the_term_by_term_form(eng, monkeypatch):
    """The 16 -> 10 kHz interpolation of the voiced intervals takes five weight sets per interval (the fractional position
    repeats every 5 outputs); RSAF_CPP_POLYPHASE=0 evaluates Praat's formula term by term for every output.  Every
    resampled sample of 12 s clips within 1e-9 of the interval's largest sample, CPPS within 1e-9."""
    import torch
    clips = [synth.synth_clip(930 + k, 12.0) for k in range(3)]
    wav, offs, lens = _pack(clips)
    gp = eng.clip_peaks(wav, offs, lens)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RSAF_CPP_POLYPHASE", mode)
        got = eng.cpp(wav, offs, lens, gp, 100.0, 500.0).cpu().numpy()
        torch.cuda.synchronize()
        L = eng._last_cpp
        hdr = L["hdr"].cpu().numpy().reshape(len(clips), 4)
        res = L["res"].cpu().numpy().reshape(len(clips), L["cap_res"])
        out[mode] = (got, [res[i, :hdr[i, 3]].copy() for i in range(len(clips))])
    monkeypatch.delenv("RSAF_CPP_POLYPHASE", raising=False)
    assert sum(len(r) for r in out["1"][1]) > 100000
    for a, b in zip(out["1"][1], out["0"][1]):
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * np.abs(b).max()
    ok = ~np.isnan(out["0"][0])
    assert np.array_equal(np.isnan(out["1"][0]), ~ok)
    assert np.abs(out["1"][0][ok] - out["0"][0][ok]).max() <= 1e-9 * np.abs(out["0"][0][ok]).max()
