"""Pin the training-step oracle against vectors captured from the reference module in training mode
(tests/golden/make_cnnlstm_train_golden.py; reference = src/models.py + src/dl_cv_strategies.py:241-243)."""
import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from weights import sample_tensor, synth_input, synth_state_dict  # noqa: E402

from oracle import cnnlstm_train_oracle as to

CASES = sorted(glob.glob(os.path.join(HERE, "golden", "cnnlstm_train_*.npz")))


def load_case(path):
    z = np.load(path)
    D, C, H, B, T, seed = [int(v) for v in z["meta"]]
    return z, D, C, H, B, T, seed, str(z["act"]), float(z["lr"])


def close(a, b, rtol, what):
    """a, b: sample_tensor vectors; compare relative to the tensor's RMS-scale (sum of squares is the last entry)."""
    scale = max(np.abs(b[:-2]).max(), 1e-30)
    err = np.abs(a[:-2] - b[:-2]).max() / scale
    assert err < rtol, (what, err)
    assert abs(a[-1] - b[-1]) <= rtol * max(abs(b[-1]), 1e-30) * 4, (what, "sum of squares", a[-1], b[-1])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[14:-4] for p in CASES])
def test_step_matches_reference(path):
    z, D, C, H, B, T, seed, act, lr = load_case(path)
    sd = synth_state_dict(D, C, H, seed)
    x = synth_input(B, T, D, seed + 1000)
    r = to.forward_backward(sd, x, z["labels"], act)
    assert np.abs(r["logits"] - z["logits"]).max() < 1e-10
    assert abs(r["loss"] - float(z["loss"])) < 1e-10
    for k, g in r["grads"].items():
        ref = z["grad/" + k]
        if k.endswith("conv1.bias") or k.endswith("conv2.bias") or k.endswith("shortcut.0.bias") or k.endswith("attention_weights.bias"):
            # mathematically zero (a bias in front of batch-norm / inside a softmax): only rounding noise on both sides
            assert np.abs(sample_tensor(g)[:-2]).max() < 1e-12 and np.abs(ref[:-2]).max() < 1e-12, k
            continue
        close(sample_tensor(g), ref, 1e-8, k)
    # BN buffers and one Adam step
    for k, v in to.updated_bn_buffers(sd, r["bn_stats"]).items():
        close(sample_tensor(v), z["after/" + k], 1e-10, k)
    params = {k: np.asarray(v, np.float64) for k, v in sd.items() if k in r["grads"]}
    to.adam_step(params, r["grads"], {}, lr)
    for k, v in params.items():
        ref = z["after/" + k]
        # the first Adam step moves every weight by lr * g / (|g| + eps): for the mathematically-zero gradients the
        # direction is rounding noise, so those tensors are only required to stay within lr of their start
        if k.endswith("conv1.bias") or k.endswith("conv2.bias") or k.endswith("shortcut.0.bias") or k.endswith("attention_weights.bias"):
            assert np.abs(sample_tensor(v)[:-2] - ref[:-2]).max() <= 2 * lr + 1e-12, k
            continue
        close(sample_tensor(v), ref, 1e-7, "after/" + k)


def test_masks_scale_and_shapes():
    mk = to.make_masks(2, 9, 8, 4, 0.2, 0.5, seed=3)
    assert mk["res_block1"].shape == (2, 9, 8) and mk["res_block2"].shape == (2, 4, 8)
    assert mk["lstm0"].shape == (2, 4, 8) and mk["fc"].shape == (2, 8)
    assert set(np.unique(mk["res_block1"])) <= {0.0, np.float32(1.25)}
    assert set(np.unique(mk["fc"])) <= {0.0, np.float32(2.0)}


def test_dropout_mask_changes_the_gradient_path():
    D, C, H, B, T = 8, 8, 64, 2, 6
    sd = synth_state_dict(D, C, H, 5)
    x = synth_input(B, T, D, 6)
    mk = to.make_masks(B, T, C, H, 0.5, 0.5, seed=9)
    a = to.forward_backward(sd, x, [0, 1], "silu")
    b = to.forward_backward(sd, x, [0, 1], "silu", masks=mk)
    assert np.abs(a["logits"] - b["logits"]).max() > 1e-6
    # the gradient w.r.t. fc.weight is dlogits^T . (ctx * mask): dropped features have exactly zero gradient
    dropped = mk["fc"] == 0
    g = b["grads"]["fc.weight"]
    both = dropped.all(axis=0)
    assert np.all(g[:, both] == 0.0)
