"""Generate tests/golden/w2v2_small.npz from the installed third-party ``transformers`` module.

The arithmetic of the reference's Wav2Vec2 path lives in transformers' Wav2Vec2Model /
Wav2Vec2FeatureExtractor (src/foundation_model_extractor.py:70-72,113-116).  This script feeds the
build's seeded random weights into THAT module (no fetch: constructed from a config) and stores
its outputs.  Run in the build container:  python tests/golden/make_w2v2_golden.py
"""
import json
import os
import sys

import numpy as np
import torch
from transformers import Wav2Vec2Config, Wav2Vec2FeatureExtractor, Wav2Vec2Model

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from robust_speech_analysis_framework_amd.w2v2_config import W2V2Config, random_state_dict  # noqa: E402
from robust_speech_analysis_framework_amd import synth  # noqa: E402

SMALL = dict(conv_dim=(32,) * 7, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
             intermediate_size=128, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)


def hf_model(cfg: W2V2Config, sd):
    hc = Wav2Vec2Config(conv_dim=cfg.conv_dim, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                        num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups)
    m = Wav2Vec2Model(hc)
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and set(res.missing_keys) <= {"masked_spec_embed"}, res
    return m.eval()


if __name__ == "__main__":
    torch.set_num_threads(4)
    cfg = W2V2Config(**SMALL)
    sd = random_state_dict(cfg, seed=7)
    m = hf_model(cfg, sd)
    fe = Wav2Vec2FeatureExtractor()
    out = {"cfg": np.array(json.dumps(SMALL))}
    clip = synth.synth_clip(50, 2.0)                      # 32 000 samples
    for n in (8000, 20000):
        x = clip[:n]
        iv = fe(x, sampling_rate=16000, return_tensors="pt").input_values
        with torch.no_grad():
            o = m(iv)
        out[f"input_values_{n}"] = iv.numpy()[0]
        out[f"extract_features_{n}"] = o.extract_features.numpy()[0]
        out[f"last_hidden_state_{n}"] = o.last_hidden_state.numpy()[0]
        print(n, o.last_hidden_state.shape, float(o.last_hidden_state.abs().max()))
    np.savez_compressed(os.path.join(HERE, "w2v2_small.npz"), **out)
    # integer contract: chunk plan + frame counts of the reference loop at the base geometry
    base = W2V2Config()
    table = {}
    for n in (7999, 8000, 64000, 71999, 72000, 80000, 144000, 480000):
        plan = []
        if n >= 8000:
            for i in range(0, n, 64000):
                ln = min(80000, n - i)
                if ln >= 8000:
                    plan.append([i, ln, base.frames(ln)])
        table[str(n)] = plan
    # cross-check frame counts with transformers' own length formula
    full = Wav2Vec2Model(Wav2Vec2Config())
    for n, plan in table.items():
        for _, ln, fr in plan:
            assert int(full._get_feat_extract_output_lengths(ln)) == fr
    with open(os.path.join(HERE, "w2v2_chunking_cases.json"), "w") as f:
        json.dump(table, f, indent=1)
    print({k: sum(p[2] for p in v) for k, v in table.items()})
