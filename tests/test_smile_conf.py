"""`validate_smile_conf`: a config whose pitch-chain / cSpectral / smoother options differ from what the kernels hard-code is
refused (reference conventions: fatal message + empty DataFrame, src/opensmile_extractor.py:41-43), not silently analysed
with the built-in values.  The config text below is this repository's own (same option names as openSMILE)."""
import pytest

from robust_speech_analysis_framework_amd import smile

_BASE = """
[fr1:cFramer]
frameSize = 0.0250
frameStep = 0.010
[pe2:cVectorPreemphasis]
k = 0.97
[w1:cWindower]
winFunc = ham
[mspec:cMelspec]
htkcompatible = 1
usePower = 0
lofreq = 20
hifreq = 8000
[mfcc:cMfcc]
firstMfcc = 1
lastMfcc = 12
[energy:cEnergy]
rms = 1
log = 0
[scale:cSpecScale]
scale = octave
sourceScale = lin
interpMethod = spline
minF = 25
maxF = -1
specSmooth = 1
specEnhance = 1
auditoryWeighting = 1 ; trailing comments are not part of the value
[shs:cPitchShs]
maxPitch = 620
minPitch = 52
nCandidates = 6
voicingCutoff = 0.700000
nHarmonics = 15
compressionFactor = 0.850000
greedyPeakAlgo = 1
[pitchSmooth:cPitchSmootherViterbi]
bufferLength=30
wTvv =10.0
wTvvd= 5.0
wTvuv=10.0
wThr = 4.0
wTuu = 0.0
wLocal=2.0
wRange=1.0
[volmerge:cValbasedSelector]
threshold=0.001
[pitchJitter:cPitchJitter]
F0field = F0final
searchRangeRel = 0.250000
onlyVoiced = 0
[spectral:cSpectral]
bands[0]=250-650
bands[1]=1000-4000
rollOff[0] = 0.25
rollOff[1] = 0.50
rollOff[2] = 0.75
rollOff[3] = 0.90
tonality = 0
[lld:cContourSmoother]
nameAppend = sma
[delta1:cDeltaRegression]
deltawin=2
[delta2:cDeltaRegression]
deltawin=2
[delta3:cDeltaRegression]
deltawin=2
[functL1:cFunctionals]
frameSize=0.025
frameStep=0
functionalsEnabled=Extremes;Regression;Moments
"""


def _write(tmp_path, text):
    p = tmp_path / "chain.conf"
    p.write_text(text)
    return str(p)


def test_the_chain_the_kernels_implement_is_accepted(tmp_path):
    sec = smile.validate_smile_conf(_write(tmp_path, _BASE))
    assert sec["scale"]["auditoryweighting"] == "1" and smile.functionals_window(sec) == 3


@pytest.mark.parametrize("old,new", [
    ("wTvv =10.0", "wTvv =12.0"), ("wTvvd= 5.0", "wTvvd= 4.0"), ("wThr = 4.0", "wThr = 3"), ("bufferLength=30", "bufferLength=40"),
    ("nCandidates = 6", "nCandidates = 5"), ("nHarmonics = 15", "nHarmonics = 12"), ("compressionFactor = 0.850000", "compressionFactor = 0.9"),
    ("voicingCutoff = 0.700000", "voicingCutoff = 0.5"), ("maxPitch = 620", "maxPitch = 500"), ("greedyPeakAlgo = 1", "greedyPeakAlgo = 0"),
    ("specEnhance = 1", "specEnhance = 0"), ("interpMethod = spline", "interpMethod = linear"), ("minF = 25", "minF = 50"),
    ("threshold=0.001", "threshold=0.01"), ("searchRangeRel = 0.250000", "searchRangeRel = 0.3"), ("onlyVoiced = 0", "onlyVoiced = 1"),
    ("rollOff[2] = 0.75", "rollOff[2] = 0.8"), ("bands[0]=250-650", "bands[0]=0-500"), ("tonality = 0", "tonality = 1"),
    ("rms = 1", "rms = 0"), ("nameAppend = sma", "nameAppend = sma\nsmaWin = 5"),
])
def test_a_changed_option_is_refused(tmp_path, old, new):
    assert old in _BASE
    with pytest.raises(ValueError) as e:
        smile.validate_smile_conf(_write(tmp_path, _BASE.replace(old, new)))
    assert "unsupported" in str(e.value)


def test_dropin_follows_the_fatal_error_convention(tmp_path, capsys):
    import pandas as pd
    out = smile.extract_opensmile_features(pd.DataFrame({"filepath": []}), "x", _write(tmp_path, _BASE.replace("wLocal=2.0", "wLocal=1.0")),
                                           verbose=False)
    assert out.empty and "FATAL ERROR" in capsys.readouterr().out
