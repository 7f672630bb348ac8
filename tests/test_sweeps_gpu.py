"""The sweeps that found this project's real defects (tests/sweeps/*), at a size the suite can afford: the regression net of
the time-axis work (x1 / xmax), the float64 openSMILE chain and the pitch kernels.  Larger runs: `python tests/sweeps/...`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mshds_fuzz_six_random_clips(rsaf_lib):
    from tests.sweeps import mshds_fuzz
    r = mshds_fuzz.run(first=6100, count=6, min_s=0.6, max_s=3.0, verbose=False, workers=6)
    assert r["mismatches"] == 0 and r["worst_rel"] <= 1e-4, r


def test_mshds_edge_seventeen_degenerate_inputs(rsaf_lib):
    from tests.sweeps import mshds_edge
    r = mshds_edge.run()
    assert r["cases"] == 17 and r["mismatches"] == 0, r


def test_stage_fuzz_four_random_clips(rsaf_lib):
    from tests.sweeps import stage_fuzz
    r = stage_fuzz.run(first=8100, count=4)
    assert r["w2v2_worst"] <= 1e-4 and r["logits_worst"] <= 1e-4, r


def test_whole_30s_wav2vec2_sequence_matches_oracle(rsaf_lib):
    """BASELINE config C3's unit of work: one full 30 s clip -> 8 windows -> 1 842 x 768 frames at base geometry, every
    value against the torch-CPU restatement (the window tests in test_w2v2_gpu.py compare 5 s / 2 s pieces)."""
    import torch
    from oracle import w2v2_oracle
    from robust_speech_analysis_framework_amd import synth
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    from robust_speech_analysis_framework_amd.w2v2_config import W2V2Config, random_state_dict
    cfg = W2V2Config()
    sd = random_state_dict(cfg, 0)
    eng = W2V2Engine(cfg, sd, torch.device("cuda:0"))
    clip = synth.synth_clip(20260400, 30.0)
    seq, fo = eng.extract_packed(torch.from_numpy(clip).cuda(), [0], [len(clip)])
    torch.cuda.synchronize()
    ref = w2v2_oracle.extract_sequence(sd, cfg, clip)
    got = seq.cpu().numpy()
    assert got.shape == ref.shape == (1842, 768) and int(fo[1]) == 1842
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    rows = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    assert rows.max() <= 1e-4, int(np.argmax(rows))                     # per frame as well, incl. the 99-frame tail window
