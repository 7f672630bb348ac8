"""CPU restatement of the Wav2Vec2 frame-embedding path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

* the chunk loop of ``extract_wav2vec2_sequences`` (``src/foundation_model_extractor.py:87-125``):
  windows of 80 000 samples every 64 000, windows < 8 000 samples dropped, per-chunk zero-mean /
  unit-variance normalisation (HF ``Wav2Vec2FeatureExtractor``, ``feature_extraction_wav2vec2.py:95``),
  batch-1 forward, ``np.vstack`` of all chunk outputs (the overlapped second is duplicated);
* ``Wav2Vec2Model.forward`` in eval mode (third-party dependency ``transformers``, pinned 4.54.1 by
  the reference, 5.15.0 installed): ``modeling_wav2vec2.py:254-323`` (conv stack, GroupNorm on
  layer 0, GELU), ``:422-434`` (LayerNorm + projection), ``:326-379`` (weight-normed grouped
  positional conv, drop last frame, GELU), ``:575-608`` (post-LN encoder layer), ``:657-726``.

Pinned by ``tests/golden/w2v2_small_*.npz`` (outputs of the installed ``transformers`` module on a
small seeded geometry, ``tests/golden/make_w2v2_golden.py``) and, in the build container, directly
against ``transformers`` at the base geometry.  Real pretrained weights are unavailable offline, so
parity on ``facebook/wav2vec2-base-960h`` values themselves is not pinned.

Arithmetic: torch CPU float32 functional ops (what the reference's CPU path executes).
"""
from __future__ import annotations

import numpy as np


def chunk_starts(n_samples, chunk_seconds=5, overlap_seconds=1, sample_rate=16000):
    """[(start, length)] following ``src/foundation_model_extractor.py:97-108`` line by line."""
    chunk_size = int(sample_rate * chunk_seconds)                       # :97
    step_size = int(sample_rate * (chunk_seconds - overlap_seconds))    # :98
    out = []
    for i in range(0, n_samples, step_size):                            # :103
        length = len(range(i, min(i + chunk_size, n_samples)))          # waveform[:, i:i+chunk_size]
        if length < int(sample_rate * 0.5):                             # :107
            continue
        out.append((i, length))
    return out


def file_is_skipped(n_samples_at_native_rate, sample_rate=16000):
    """``:88``: files shorter than 0.5 s (tested on the pre-resample sample count) are skipped."""
    return n_samples_at_native_rate < int(sample_rate * 0.5)


def feat_lengths(n, kernels=(10, 3, 3, 3, 3, 2, 2), strides=(5, 2, 2, 2, 2, 2, 2)):
    for k, s in zip(kernels, strides):
        n = (n - k) // s + 1
    return n


def hf_normalize(x):
    """``Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm`` without attention mask (float32 numpy)."""
    x = np.asarray(x, dtype=np.float32)
    return (x - x.mean()) / np.sqrt(x.var() + 1e-7)


def forward(sd, cfg, input_values, return_stages=False):
    """``Wav2Vec2Model(...).eval()(input_values).last_hidden_state`` restated with functional ops.

    sd: HF-keyed dict of numpy arrays; cfg: object with the HF geometry attributes;
    input_values: float32 [B, n] (already normalised)."""
    import torch
    import torch.nn.functional as F
    t = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in sd.items()}
    eps = cfg.layer_norm_eps
    st = {}
    with torch.no_grad():
        h = torch.as_tensor(np.asarray(input_values), dtype=torch.float32)[:, None]       # [B,1,n]
        for i, s in enumerate(cfg.conv_stride):
            h = F.conv1d(h, t[f"feature_extractor.conv_layers.{i}.conv.weight"], stride=s)
            if i == 0:
                C0 = h.shape[1]
                h = F.group_norm(h, C0, t["feature_extractor.conv_layers.0.layer_norm.weight"],
                                 t["feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
            h = F.gelu(h)
            if i == 0:
                st["conv0"] = h.transpose(1, 2)
        feats = h.transpose(1, 2)                                                          # [B,T,C]
        st["extract_features"] = feats
        x = F.layer_norm(feats, (feats.shape[-1],), t["feature_projection.layer_norm.weight"],
                         t["feature_projection.layer_norm.bias"], eps)
        st["extract_features_ln"] = x          # what HF returns as `.extract_features` (:1353-1354)
        x = F.linear(x, t["feature_projection.projection.weight"], t["feature_projection.projection.bias"])
        st["projected"] = x
        # positional conv embedding with weight norm over dim 2 (w = g * v / ||v||_{dims 0,1})
        g = t["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
        v = t["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
        w = g * v / v.norm(dim=(0, 1), keepdim=True)
        K = cfg.num_conv_pos_embeddings
        pos = F.conv1d(x.transpose(1, 2), w, t["encoder.pos_conv_embed.conv.bias"], padding=K // 2,
                       groups=cfg.num_conv_pos_embedding_groups)
        if K % 2 == 0:
            pos = pos[:, :, :-1]
        pos = F.gelu(pos).transpose(1, 2)
        x = F.layer_norm(x + pos, (x.shape[-1],), t["encoder.layer_norm.weight"], t["encoder.layer_norm.bias"], eps)
        st["encoder_in"] = x
        nh = cfg.num_attention_heads
        hd = x.shape[-1] // nh
        for l in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{l}."
            B, T, D = x.shape
            q = F.linear(x, t[p + "attention.q_proj.weight"], t[p + "attention.q_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
            k = F.linear(x, t[p + "attention.k_proj.weight"], t[p + "attention.k_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
            vv = F.linear(x, t[p + "attention.v_proj.weight"], t[p + "attention.v_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
            a = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1) @ vv
            a = a.transpose(1, 2).reshape(B, T, D)
            a = F.linear(a, t[p + "attention.out_proj.weight"], t[p + "attention.out_proj.bias"])
            x = F.layer_norm(x + a, (D,), t[p + "layer_norm.weight"], t[p + "layer_norm.bias"], eps)
            f = F.gelu(F.linear(x, t[p + "feed_forward.intermediate_dense.weight"], t[p + "feed_forward.intermediate_dense.bias"]))
            f = F.linear(f, t[p + "feed_forward.output_dense.weight"], t[p + "feed_forward.output_dense.bias"])
            x = F.layer_norm(x + f, (D,), t[p + "final_layer_norm.weight"], t[p + "final_layer_norm.bias"], eps)
            if l == 0:
                st["layer0"] = x
        st["last_hidden_state"] = x
    if return_stages:
        return x.numpy(), {k: v.numpy() for k, v in st.items()}
    return x.numpy()


def extract_sequence(sd, cfg, wav, chunk_seconds=5, overlap_seconds=1):
    """One clip -> float32 [T, hidden] or None (file skipped), following ``:87-125``."""
    wav = np.asarray(wav, dtype=np.float32)
    if file_is_skipped(len(wav)):
        return None
    outs = []
    for start, length in chunk_starts(len(wav), chunk_seconds, overlap_seconds):
        chunk = wav[start:start + length]
        iv = hf_normalize(chunk)[None]
        outs.append(forward(sd, cfg, iv)[0])
    if not outs:
        return None
    return np.vstack(outs)                                               # :124


def extract_embedding(seq):
    """``extract_wav2vec2_embeddings`` (``:133-166``): time mean -> dim_0..dim_{H-1}."""
    return np.mean(seq, axis=0)
