"""Deterministic synthetic CNN-LSTM state_dicts shared by the golden generator and the tests.

The generator loads these arrays into the REFERENCE module (``load_state_dict``) to capture golden
outputs; the tests regenerate the identical arrays (numpy PCG64, no torch RNG involved) to feed
the oracle and the HIP path.  Keys/shapes follow SURVEY.md App. D.
"""
import numpy as np


def cnnlstm_shapes(input_dim, C, H, num_classes=2, layers=2):
    sh = {}
    for blk, cin in (("res_block1", input_dim), ("res_block2", C)):
        sh[f"{blk}.conv1.weight"] = (C, cin, 3)
        sh[f"{blk}.conv1.bias"] = (C,)
        sh[f"{blk}.conv2.weight"] = (C, C, 3)
        sh[f"{blk}.conv2.bias"] = (C,)
        for bn in ("bn1", "bn2"):
            for p in ("weight", "bias", "running_mean", "running_var"):
                sh[f"{blk}.{bn}.{p}"] = (C,)
        if cin != C:
            sh[f"{blk}.shortcut.0.weight"] = (C, cin, 1)
            sh[f"{blk}.shortcut.0.bias"] = (C,)
            for p in ("weight", "bias", "running_mean", "running_var"):
                sh[f"{blk}.shortcut.1.{p}"] = (C,)
    for l in range(layers):
        inp = C if l == 0 else 2 * H
        for sfx in ("", "_reverse"):
            sh[f"lstm.weight_ih_l{l}{sfx}"] = (4 * H, inp)
            sh[f"lstm.weight_hh_l{l}{sfx}"] = (4 * H, H)
            sh[f"lstm.bias_ih_l{l}{sfx}"] = (4 * H,)
            sh[f"lstm.bias_hh_l{l}{sfx}"] = (4 * H,)
    sh["attention_pooling.attention_weights.weight"] = (1, 2 * H)
    sh["attention_pooling.attention_weights.bias"] = (1,)
    sh["fc.weight"] = (num_classes, 2 * H)
    sh["fc.bias"] = (num_classes,)
    return sh


def synth_state_dict(input_dim, C, H, seed, num_classes=2, layers=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for k, shape in cnnlstm_shapes(input_dim, C, H, num_classes, layers).items():
        if k.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif k.endswith("running_mean"):
            v = 0.2 * rng.standard_normal(shape)
        elif ".bn" in k or "shortcut.1" in k:
            v = (1.0 + 0.1 * rng.standard_normal(shape)) if k.endswith("weight") else 0.1 * rng.standard_normal(shape)
        elif k.endswith("bias"):
            v = 0.1 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) / np.sqrt(fan_in)
            if k.startswith("attention_pooling"):
                v = v * 4.0                       # make the softmax over time non-trivial
        sd[k] = v.astype(np.float32)
    return sd


def synth_input(B, T, D, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((B, T, D)).astype(np.float32)


def sample_tensor(a, cap=512):
    """Fixture-sized view of a tensor: (evenly spaced samples of the flattened array, sum, sum of squares)."""
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, f.size - 1, min(cap, f.size)).astype(np.int64)
    return np.concatenate([f[idx], [f.sum(), (f * f).sum()]])
