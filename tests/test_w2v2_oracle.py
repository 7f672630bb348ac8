"""Pin the Wav2Vec2 oracle: golden vectors from the installed ``transformers`` (small geometry),
the integer chunking contract, and (build container only) the base geometry directly."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import w2v2_oracle as wo
from robust_speech_analysis_framework_amd import synth
from robust_speech_analysis_framework_amd.w2v2_config import W2V2Config, chunk_plan, random_state_dict


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _small():
    z = np.load(os.path.join(HERE, "golden", "w2v2_small.npz"))
    cfg = W2V2Config(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(z["cfg"])).items()})
    return z, cfg, random_state_dict(cfg, seed=7)


@pytest.mark.parametrize("n", [8000, 20000])
def test_oracle_matches_transformers_golden(n):
    z, cfg, sd = _small()
    clip = synth.synth_clip(50, 2.0)[:n]
    iv = wo.hf_normalize(clip)
    assert _rel(iv, z[f"input_values_{n}"]) < 1e-6                       # HF feature-extractor normaliser
    out, st = wo.forward(sd, cfg, z[f"input_values_{n}"][None], return_stages=True)
    assert st["extract_features_ln"][0].shape == z[f"extract_features_{n}"].shape
    assert _rel(st["extract_features_ln"][0], z[f"extract_features_{n}"]) < 5e-6
    assert _rel(out[0], z[f"last_hidden_state_{n}"]) < 5e-6


def test_chunking_contract_integer_exact():
    table = json.load(open(os.path.join(HERE, "golden", "w2v2_chunking_cases.json")))
    base = W2V2Config()
    for n, plan in table.items():
        n = int(n)
        want = [(s, l) for s, l, _ in plan]
        got_oracle = [] if wo.file_is_skipped(n) else wo.chunk_starts(n)
        got_product = chunk_plan(n) if n >= 8000 else []
        assert got_oracle == want and got_product == want, n
        assert [wo.feat_lengths(l) for _, l in want] == [f for _, _, f in plan]
        assert [base.frames(l) for _, l in want] == [f for _, _, f in plan]
    assert sum(f for _, _, f in table["480000"]) == 1842 and sum(f for _, _, f in table["80000"]) == 298


def test_extract_sequence_concatenates_chunks_with_duplicated_overlap():
    z, cfg, sd = _small()
    wav = synth.synth_clip(51, 5.5)                                      # 88 000 samples -> 2 chunks
    seq = wo.extract_sequence(sd, cfg, wav)
    plan = wo.chunk_starts(len(wav))
    assert plan == [(0, 80000), (64000, 24000)]
    assert seq.shape == (cfg.frames(80000) + cfg.frames(24000), cfg.hidden_size)
    tail = wo.forward(sd, cfg, wo.hf_normalize(wav[64000:])[None])[0]
    assert np.array_equal(seq[cfg.frames(80000):], tail)                 # per-chunk normalisation + vstack
    assert wo.extract_sequence(sd, cfg, wav[:7999]) is None
    assert wo.extract_embedding(seq).shape == (cfg.hidden_size,)


@pytest.mark.skipif(os.environ.get("RSAF_SLOW") != "1", reason="base geometry vs transformers: set RSAF_SLOW=1")
def test_oracle_matches_transformers_base_geometry():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import torch
    from make_w2v2_golden import hf_model
    cfg = W2V2Config()
    sd = random_state_dict(cfg, seed=0)
    iv = wo.hf_normalize(synth.synth_clip(52, 1.0))[None]
    with torch.no_grad():
        ref = hf_model(cfg, sd)(torch.from_numpy(iv)).last_hidden_state.numpy()
    assert _rel(wo.forward(sd, cfg, iv), ref) < 1e-5
