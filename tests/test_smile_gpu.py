"""HIP openSMILE-style chain vs the CPU oracle (called through the C ABI)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import smile_oracle as so

# parity metric of SURVEY.md §8d: per LLD row / feature column, max|gpu-cpu| / max|cpu|
TOL = 1e-4
ROLLOFF_ROWS = (24, 25, 26, 27)
# spectralFlatness = exp(mean(log P)) / mean(P): the log of the weakest bins (pre-emphasised
# low-frequency bins ~1e-11 in power) is dominated by float32 FFT rounding (~6e-8 * |X|max per
# bin), so this one row is compared at 1e-3 (openSMILE itself is float32, FLOAT_DMEM).
ROW_TOL = {37: 1e-3}


def _clips(seconds_list, first=0):
    from robust_speech_analysis_framework_amd import synth
    return [synth.synth_clip(first + i, s) for i, s in enumerate(seconds_list)]


def _lld_gpu(clips):
    import torch
    from robust_speech_analysis_framework_amd import smile
    p = smile.pack_clips(clips)
    lld = smile.smile_lld(p)
    torch.cuda.synchronize()
    return p, lld


def _check_lld(gpu, ref):
    built = [i for i in range(so.NLLD) if i not in so.LLD_NOT_BUILT]
    assert np.isnan(gpu[so.LLD_NOT_BUILT]).all()
    assert not np.isnan(gpu[built]).any()
    for i in built:
        scale = np.max(np.abs(ref[i])) + 1e-30
        err = np.abs(gpu[i] - ref[i])
        if i in ROLLOFF_ROWS:
            # a roll-off is a bin index: fp32 vs fp64 cumulative sums may flip a threshold crossing
            # by one bin on isolated frames; everything else must be exact
            assert (err > 1e-3).mean() <= 2e-3, (so.LLD_NAMES[i], (err > 1e-3).mean())
            assert err.max() <= so.DF + 1e-3
        else:
            assert err.max() / scale <= ROW_TOL.get(i, TOL), (so.LLD_NAMES[i], err.max() / scale)


def test_lld_parity_ragged_batch(rsaf_lib):
    clips = _clips([5.0, 1.003, 0.025, 2.51, 0.0349])
    p, lld = _lld_gpu(clips)
    assert p.frames == [so.n_frames(len(c)) for c in clips]       # integer-exact frame counts
    g = lld.cpu().numpy().astype(np.float64)
    ref = np.concatenate([so.lld(c) for c in clips], axis=1)
    assert g.shape == ref.shape
    _check_lld(g, ref)


def test_lld_parity_30s_clip(rsaf_lib):
    clips = _clips([30.0], first=7)
    p, lld = _lld_gpu(clips)
    assert p.frames == [2998]
    _check_lld(lld.cpu().numpy().astype(np.float64), so.lld(clips[0]))


def test_empty_and_too_short_inputs(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd import smile
    p = smile.pack_clips([np.zeros(0, np.float32), np.zeros(399, np.float32)])
    assert p.frames == [0, 0]
    f = smile.smile_features(p)
    torch.cuda.synchronize()
    assert f.shape == (2, 912) and torch.isnan(f).all()
    p0 = smile.pack_clips([])
    assert smile.smile_features(p0).shape == (0, 912)


def test_functionals_exact_on_identical_input(rsaf_lib):
    """Functionals kernel vs oracle on the SAME (GPU-produced) LLD: positions bit-exact."""
    import torch
    from robust_speech_analysis_framework_amd import smile
    clips = _clips([5.0, 3.2, 0.05, 7.7], first=20)
    p, lld = _lld_gpu(clips)
    f = smile.smile_functionals(lld, p)
    torch.cuda.synchronize()
    g = f.cpu().numpy().astype(np.float64)
    L = lld.cpu().numpy().astype(np.float64)
    names = so.feature_names()
    pos = np.array([n.endswith("Pos") for n in names])
    off = 0
    refs = []
    for nf in p.frames:
        # the kernel forms sma3/delta in float32 (openSMILE is FLOAT_DMEM); mirror that so the only
        # difference left is the float64 accumulation order of the statistics themselves
        x32 = lld.cpu().numpy()[:, off:off + nf]
        if nf == 0:
            refs.append(np.full(912, np.nan))
        else:
            s32 = _sma32(x32).astype(np.float64)
            d32 = _delta32(_sma32(x32)).astype(np.float64)
            with np.errstate(invalid="ignore"):
                fs, fd = so.functionals12(s32), so.functionals12(d32)
            bad = np.isnan(x32).any(axis=1)
            fs[bad] = np.nan
            fd[bad] = np.nan
            parts = []
            for lo, hi in so.LEVELS:
                parts += [fs[lo:hi].reshape(-1), fd[lo:hi].reshape(-1)]
            refs.append(np.concatenate(parts))
        off += nf
    ref = np.stack(refs)
    nan_ref = np.isnan(ref)
    assert (np.isnan(g) == nan_ref).all()
    ok = ~nan_ref
    scale = np.max(np.abs(np.where(ok, ref, 0.0)), axis=0) + 1e-30
    err = np.abs(np.where(ok, g - ref, 0.0)).max(axis=0) / scale
    # positions: exact unless the float32 sma/delta rounding creates a tie the float64 oracle
    # does not see; verify against a float32 restatement of sma/delta instead
    assert err[~pos].max() <= TOL, (names[int(np.argmax(np.where(~pos, err, 0)))], err[~pos].max())
    off = 0
    for ci, nf in enumerate(p.frames):
        if nf == 0:
            continue
        x32 = lld.cpu().numpy()[:, off:off + nf]
        s32 = _sma32(x32)
        d32 = _delta32(s32)
        built = [i for i in range(so.NLLD) if i not in so.LLD_NOT_BUILT]
        for i in built:
            for contour, tag in ((s32[i], "_sma"), (d32[i], "_sma_de")):
                base = names.index(_col(i, tag, "maxPos"))
                assert g[ci, base] == float(np.argmax(contour)), (ci, i, tag)
                assert g[ci, base + 1] == float(np.argmin(contour)), (ci, i, tag)
        off += nf


def _col(i, tag, fn):
    n = so.LLD_NAMES[i]
    base = ("mfcc" + tag + n[4:]) if n.startswith("mfcc[") else (n + tag)
    return f"{base}_{fn}"


def _sma32(x):
    p = np.pad(x, ((0, 0), (1, 1)), mode="edge").astype(np.float32)
    return ((p[:, :-2] + p[:, 1:-1]) + p[:, 2:]) / np.float32(3.0)


def _delta32(s):
    p = np.pad(s, ((0, 0), (2, 2)), mode="edge").astype(np.float32)
    return ((p[:, 3:-1] - p[:, 1:-3]) + np.float32(2.0) * (p[:, 4:] - p[:, :-4])) / np.float32(10.0)


def test_end_to_end_features_parity(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd import smile
    clips = _clips([5.0] * 4, first=40)
    p = smile.pack_clips(np.stack(clips))
    f = smile.smile_features(p)
    torch.cuda.synchronize()
    g = f.cpu().numpy().astype(np.float64)
    ref = np.stack([so.extract(c) for c in clips])
    names = so.feature_names()
    ok = ~np.isnan(ref)
    assert (np.isnan(g) == ~ok).all()
    scale = np.max(np.abs(np.where(ok, ref, 0.0)), axis=0) + 1e-30
    err = np.abs(np.where(ok, g - ref, 0.0)).max(axis=0) / scale
    # discrete outputs (positions, roll-off extrema) can move by one step under fp32-vs-fp64
    # rounding; they are pinned exactly in test_functionals_exact_on_identical_input
    discrete = np.array([n.endswith("Pos") or "RollOff" in n for n in names])
    worst = int(np.argmax(np.where(discrete, 0, err)))
    assert err[~discrete].max() <= 5e-4, (names[worst], err[worst])


def test_dropin_dataframe_contract(rsaf_lib, tmp_path):
    import pandas as pd
    from robust_speech_analysis_framework_amd import smile, synth
    paths = synth.write_synth_corpus(str(tmp_path), 3, 1.5)
    bad = tmp_path / "broken.wav"
    bad.write_bytes(b"not a wav")
    conf = tmp_path / "Androids.conf"
    conf.write_text(_MINI_CONF)
    df = pd.DataFrame({"filepath": [paths[0], str(bad), paths[1], paths[2]]})
    out = smile.extract_opensmile_features(df, "/nonexistent/SMILExtract", str(conf), verbose=False)
    assert list(out.columns) == so.feature_names() + ["filename"]
    assert list(out["filename"]) == ["synth_00000.wav", "synth_00001.wav", "synth_00002.wav"]
    empty = smile.extract_opensmile_features(df, "x", str(tmp_path / "missing.conf"), verbose=False)
    assert empty.empty


_MINI_CONF = """
[fr1:cFramer]
frameSize=0.0250
frameStep = 0.010
[pe2:cVectorPreemphasis]
k=0.97
[w1:cWindower]
winFunc = ham
[mspec:cMelspec]
htkcompatible = 1
usePower = 0
lofreq = 20
hifreq = 8000
[mfcc:cMfcc]
firstMfcc = 1
lastMfcc =  12
[delta1:cDeltaRegression]
deltawin=2
[delta2:cDeltaRegression]
deltawin=2
[delta3:cDeltaRegression]
deltawin=2
[functL1:cFunctionals]
functionalsEnabled=Extremes;Regression;Moments
"""
