"""Pin the CNN-LSTM oracle against golden vectors captured from the reference module
(tests/golden/make_cnnlstm_golden.py; reference = src/models.py:161-193)."""
import glob
import hashlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from weights import synth_input, synth_state_dict  # noqa: E402

from oracle import cnnlstm_oracle as co

CASES = sorted(glob.glob(os.path.join(HERE, "golden", "cnnlstm_d*.npz")))


def _load(path):
    z = np.load(path)
    D, C, H, B, T, seed = [int(v) for v in z["meta"]]
    return z, D, C, H, B, T, seed, str(z["act"])


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_torch_restatement_matches_reference(path):
    z, D, C, H, B, T, seed, act = _load(path)
    sd = synth_state_dict(D, C, H, seed)
    x = synth_input(B, T, D, seed + 1000)
    logits, st = co.forward_torch(sd, x, act, return_stages=True)
    for k in z.files:
        if k in ("meta", "act"):
            continue
        assert st[k].shape == z[k].shape
        assert _rel(st[k], z[k]) < 2e-6, (k, _rel(st[k], z[k]))


@pytest.mark.parametrize("path", [p for p in CASES if "_d16_" in p],
                         ids=[os.path.basename(p)[8:-4] for p in CASES if "_d16_" in p])
def test_first_principles_numpy_matches_reference(path):
    z, D, C, H, B, T, seed, act = _load(path)
    sd = synth_state_dict(D, C, H, seed)
    x = synth_input(B, T, D, seed + 1000)
    logits, st = co.forward_numpy(sd, x, act, return_stages=True)
    for k in ("res1", "res2", "lstm", "pooled", "logits"):
        assert _rel(st[k], z[k]) < 1e-5, (k, _rel(st[k], z[k]))
    assert st["pool"].shape[1] == T // 2                      # floor pooling (odd T)


def test_zero_padding_without_mask_changes_logits():
    """collate_fn pads with zeros and nothing masks them (src/dl_cv_strategies.py:81-84): the padded
    sample's logits differ from the same sample alone, and the oracle reproduces both."""
    z = np.load(os.path.join(HERE, "golden", "cnnlstm_ragged_pad.npz"))
    D, C, H, B, T, seed = [int(v) for v in z["meta"]]
    sd = synth_state_dict(D, C, H, seed)
    a, b = synth_input(1, 37, D, 2001)[0], synth_input(1, 64, D, 2002)[0]
    xp = co.collate_zero_pad([a, b])
    assert xp.shape == (2, 64, D) and (xp[0, 37:] == 0).all()
    assert _rel(co.forward_torch(sd, xp), z["logits_padded"]) < 2e-6
    assert _rel(co.forward_torch(sd, a[None]), z["logits_alone"]) < 2e-6
    assert np.abs(z["logits_padded"][0] - z["logits_alone"][0]).max() > 1e-3
    prob, pred = co.eval_outputs(z["logits_padded"])
    assert prob.shape == (2,) and pred.dtype == np.int64 and ((prob > 0.5) == (pred == 1)).all()


def test_shipped_reading_checkpoint_logits():
    z = np.load(os.path.join(HERE, "golden", "cnnlstm_shipped_ckpt_logits.npz"))
    sd = dict(np.load(os.path.join(HERE, "golden", "cnnlstm_ckpt_reading_state.npz")))
    x = synth_input(2, 300, 768, 3000)
    got = co.forward_torch(sd, x, str(z["reading_act"]))
    assert _rel(got, z["reading_logits"]) < 2e-6


@pytest.mark.skipif(not os.path.exists("/root/reference/models/final_tuned_cnn_lstm_combined.pt"),
                    reason="shipped checkpoint only exists in the build container")
def test_shipped_combined_checkpoint_logits_in_build_container():
    import torch
    path = "/root/reference/models/final_tuned_cnn_lstm_combined.pt"
    z = np.load(os.path.join(HERE, "golden", "cnnlstm_shipped_ckpt_logits.npz"))
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == str(z["combined_sha256"])
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = {k: v.numpy() for k, v in ck["model_state_dict"].items()}
    got = co.forward_torch(sd, synth_input(2, 300, 768, 3000), str(z["combined_act"]))
    assert _rel(got, z["combined_logits"]) < 2e-6


def test_unknown_activation_raises_value_error():
    sd = synth_state_dict(16, 32, 64, 1)
    with pytest.raises(ValueError):
        co.forward_torch(sd, synth_input(1, 8, 16, 2), "relu")
