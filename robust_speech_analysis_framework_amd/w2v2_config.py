"""Wav2Vec2 geometry + weights for the HIP frame-embedding extractor.

The reference loads ``facebook/wav2vec2-base-960h`` by NAME through ``transformers``
(``src/foundation_model_extractor.py:70-72``).  This build never fetches: ``model_name`` must be a
local directory holding ``config.json`` and ``model.safetensors`` (HF layout), or the caller asks
for seeded random weights of the base geometry (benchmarks / parity tests).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np


@dataclass
class W2V2Config:
    conv_dim: tuple = (512,) * 7
    conv_kernel: tuple = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: tuple = (5, 2, 2, 2, 2, 2, 2)
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    extras: dict = field(default_factory=dict)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    def frames(self, n_samples: int) -> int:
        """Feature-encoder output length (transformers ``_get_feat_extract_output_lengths``)."""
        n = int(n_samples)
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
            if n <= 0:
                return 0
        return n

    def validate(self):
        c = self.conv_dim
        if len(c) != 7 or len(set(c)) != 1 or tuple(self.conv_kernel) != (10, 3, 3, 3, 3, 2, 2) \
                or tuple(self.conv_stride) != (5, 2, 2, 2, 2, 2, 2):
            raise ValueError("only the wav2vec2 7-layer feature encoder (k 10,3,3,3,3,2,2 / s 5,2,2,2,2,2,2, "
                             "uniform width) is implemented")
        if c[0] % 32 or self.hidden_size % 4 or self.intermediate_size % 4:
            raise ValueError("conv_dim must be a multiple of 32; hidden/intermediate multiples of 4")
        if self.hidden_size % self.num_attention_heads or self.head_dim % 4:
            raise ValueError("head_dim must be a multiple of 4")
        g = self.num_conv_pos_embedding_groups
        if self.hidden_size % g or (self.hidden_size // g) % 4 or self.num_conv_pos_embeddings % 2:
            raise ValueError("pos-conv: channels per group must be a multiple of 4, kernel even")

    @staticmethod
    def from_hf_dict(d: dict) -> "W2V2Config":
        for key, want in (("feat_extract_norm", "group"), ("feat_extract_activation", "gelu"),
                          ("hidden_act", "gelu"), ("do_stable_layer_norm", False), ("conv_bias", False)):
            if d.get(key, want) != want:
                raise ValueError(f"config.json: {key}={d.get(key)!r} is not supported (need {want!r})")
        return W2V2Config(conv_dim=tuple(d["conv_dim"]), conv_kernel=tuple(d["conv_kernel"]),
                          conv_stride=tuple(d["conv_stride"]), hidden_size=d["hidden_size"],
                          num_hidden_layers=d["num_hidden_layers"], num_attention_heads=d["num_attention_heads"],
                          intermediate_size=d["intermediate_size"],
                          num_conv_pos_embeddings=d["num_conv_pos_embeddings"],
                          num_conv_pos_embedding_groups=d["num_conv_pos_embedding_groups"],
                          layer_norm_eps=d.get("layer_norm_eps", 1e-5))


def hf_shapes(cfg: W2V2Config) -> dict:
    """state_dict keys/shapes of ``transformers.Wav2Vec2Model`` for this geometry."""
    sh = {}
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        sh[f"feature_extractor.conv_layers.{i}.conv.weight"] = (c, cin, k)
        cin = c
    sh["feature_extractor.conv_layers.0.layer_norm.weight"] = (cfg.conv_dim[0],)
    sh["feature_extractor.conv_layers.0.layer_norm.bias"] = (cfg.conv_dim[0],)
    Hd, Cc = cfg.hidden_size, cfg.conv_dim[-1]
    sh["feature_projection.layer_norm.weight"] = (Cc,)
    sh["feature_projection.layer_norm.bias"] = (Cc,)
    sh["feature_projection.projection.weight"] = (Hd, Cc)
    sh["feature_projection.projection.bias"] = (Hd,)
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    sh["encoder.pos_conv_embed.conv.bias"] = (Hd,)
    sh["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, K)
    sh["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (Hd, Hd // G, K)
    sh["encoder.layer_norm.weight"] = (Hd,)
    sh["encoder.layer_norm.bias"] = (Hd,)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sh[p + f"attention.{n}.weight"] = (Hd, Hd)
            sh[p + f"attention.{n}.bias"] = (Hd,)
        sh[p + "layer_norm.weight"] = (Hd,)
        sh[p + "layer_norm.bias"] = (Hd,)
        sh[p + "feed_forward.intermediate_dense.weight"] = (cfg.intermediate_size, Hd)
        sh[p + "feed_forward.intermediate_dense.bias"] = (cfg.intermediate_size,)
        sh[p + "feed_forward.output_dense.weight"] = (Hd, cfg.intermediate_size)
        sh[p + "feed_forward.output_dense.bias"] = (Hd,)
        sh[p + "final_layer_norm.weight"] = (Hd,)
        sh[p + "final_layer_norm.bias"] = (Hd,)
    return sh


def random_state_dict(cfg: W2V2Config, seed: int = 0) -> dict:
    """Seeded random weights (numpy PCG64; no torch RNG) in the HF key layout.  Scales keep every
    stage O(1) so parity errors are not hidden by vanishing activations."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for k, shape in hf_shapes(cfg).items():
        if k.endswith("layer_norm.weight"):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif k.endswith("bias"):
            v = 0.05 * rng.standard_normal(shape)
        elif k.endswith("original0"):
            v = 1.0 + 0.2 * rng.random(shape)
        elif k.endswith("original1"):
            v = rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = np.sqrt(2.0) if "conv_layers" in k else 1.0
            v = gain * rng.standard_normal(shape) / np.sqrt(fan_in)
        sd[k] = v.astype(np.float32)
    return sd


def _strip_prefix(sd: dict) -> dict:
    """base-960h is a Wav2Vec2ForCTC checkpoint: keys carry a ``wav2vec2.`` prefix; ``lm_head`` is unused."""
    out = {}
    for k, v in sd.items():
        if k.startswith("wav2vec2."):
            k = k[len("wav2vec2."):]
        if k.startswith("lm_head") or k == "masked_spec_embed":
            continue
        k = k.replace("pos_conv_embed.conv.weight_g", "pos_conv_embed.conv.parametrizations.weight.original0")
        k = k.replace("pos_conv_embed.conv.weight_v", "pos_conv_embed.conv.parametrizations.weight.original1")
        out[k] = np.asarray(v, dtype=np.float32)
    return out


def load_local_model(model_dir: str):
    """(config, state_dict) from a LOCAL HF directory.  Raises for anything that is not a local path."""
    if not os.path.isdir(model_dir):
        raise FileNotFoundError(
            f"'{model_dir}' is not a local directory; this build never downloads models "
            "(pass a directory with config.json + model.safetensors)")
    with open(os.path.join(model_dir, "config.json")) as f:
        cfg = W2V2Config.from_hf_dict(json.load(f))
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        sd = load_file(st)
    else:
        import torch
        sd = {k: v.numpy() for k, v in torch.load(os.path.join(model_dir, "pytorch_model.bin"),
                                                  map_location="cpu", weights_only=True).items()}
    sd = _strip_prefix(sd)
    missing = set(hf_shapes(cfg)) - set(sd)
    if missing:
        raise KeyError(f"checkpoint lacks {sorted(missing)[:4]} ...")
    return cfg, sd


def save_local_model(model_dir: str, cfg: W2V2Config, sd: dict):
    """Write config.json + model.safetensors (used by tests to exercise the local-directory loader)."""
    from safetensors.numpy import save_file
    os.makedirs(model_dir, exist_ok=True)
    d = {"conv_dim": list(cfg.conv_dim), "conv_kernel": list(cfg.conv_kernel), "conv_stride": list(cfg.conv_stride),
         "hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_hidden_layers,
         "num_attention_heads": cfg.num_attention_heads, "intermediate_size": cfg.intermediate_size,
         "num_conv_pos_embeddings": cfg.num_conv_pos_embeddings,
         "num_conv_pos_embedding_groups": cfg.num_conv_pos_embedding_groups,
         "layer_norm_eps": cfg.layer_norm_eps, "feat_extract_norm": "group", "feat_extract_activation": "gelu",
         "hidden_act": "gelu", "do_stable_layer_norm": False, "conv_bias": False, "model_type": "wav2vec2"}
    with open(os.path.join(model_dir, "config.json"), "w") as f:
        json.dump(d, f)
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, os.path.join(model_dir, "model.safetensors"))


# ---- chunking contract of the reference (integer-exact) -------------------------------------------
SAMPLE_RATE = 16000


def chunk_plan(n_samples: int, chunk_seconds=5, overlap_seconds=1, sample_rate: int = SAMPLE_RATE):
    """[(start, length)] exactly as ``src/foundation_model_extractor.py:97-108``: windows of
    ``chunk_seconds`` every ``chunk_seconds - overlap_seconds``; a window shorter than 0.5 s is dropped."""
    chunk = int(sample_rate * chunk_seconds)
    step = int(sample_rate * (chunk_seconds - overlap_seconds))
    if step <= 0:
        raise ValueError("range() arg 3 must not be zero" if step == 0 else "step must be positive")
    min_len = int(sample_rate * 0.5)
    plan = []
    for i in range(0, int(n_samples), step):
        ln = min(chunk, int(n_samples) - i)
        if ln < min_len:
            continue
        plan.append((i, ln))
    return plan
