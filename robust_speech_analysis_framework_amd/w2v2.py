"""Wav2Vec2 frame-embedding extractor on the HIP path (drop-in for
``src/foundation_model_extractor.py``).

The reference walks each file in 5 s windows every 4 s, normalises every window on its own, runs
``Wav2Vec2Model`` at batch 1 and stacks the window outputs (``:97-125``).  Here all windows of all
clips of a batch are planned with the same integer arithmetic, grouped by length and pushed through
``rsaf_w2v2_forward`` in large sub-batches; the final LayerNorm writes each window's frames straight
to its ``np.vstack`` position, so the values per window are those of the batch-1 reference.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .w2v2_config import SAMPLE_RATE, W2V2Config, chunk_plan, load_local_model, random_state_dict
from .wavio import read_wav_mono_device

_KERNELS = (10, 3, 3, 3, 3, 2, 2)


def _cfg_args(cfg: W2V2Config):
    return (cfg.conv_dim[0], cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
            cfg.intermediate_size, cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups)


def weight_offsets(cfg: W2V2Config):
    lib = _lib.load()
    cap = 32 + 12 * cfg.num_hidden_layers
    buf = (C.c_int64 * cap)()
    n = C.c_int(0)
    _lib.check(lib.rsaf_w2v2_weight_offsets(*_cfg_args(cfg), buf, cap, C.byref(n)), "rsaf_w2v2_weight_offsets")
    return [int(buf[i]) for i in range(n.value)], int(lib.rsaf_w2v2_weight_floats(*_cfg_args(cfg)))


def pack_weights(cfg: W2V2Config, sd: dict) -> np.ndarray:
    """HF-keyed state_dict -> the float32 blob of ``rsaf_w2v2_forward`` (weight norm folded,
    conv kernels tap-major, q/k/v fused)."""
    cfg.validate()
    offs, total = weight_offsets(cfg)
    blob = np.zeros(total, dtype=np.float32)
    it = iter(offs)

    def put(a):
        o = next(it)
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        blob[o:o + a.size] = a.astype(np.float32)

    g = lambda k: np.asarray(sd[k], dtype=np.float64)                               # noqa: E731
    put(g("feature_extractor.conv_layers.0.conv.weight")[:, 0, :])
    put(g("feature_extractor.conv_layers.0.layer_norm.weight"))
    put(g("feature_extractor.conv_layers.0.layer_norm.bias"))
    for i in range(1, 7):
        w = g(f"feature_extractor.conv_layers.{i}.conv.weight")                  # [Cout, Cin, k]
        put(np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(w.shape[0], -1))
    put(g("feature_projection.layer_norm.weight")); put(g("feature_projection.layer_norm.bias"))
    put(g("feature_projection.projection.weight")); put(g("feature_projection.projection.bias"))
    wg = g("encoder.pos_conv_embed.conv.parametrizations.weight.original0")      # [1,1,K]
    wv = g("encoder.pos_conv_embed.conv.parametrizations.weight.original1")      # [H, H/G, K]
    w = wg * wv / np.sqrt((wv * wv).sum(axis=(0, 1), keepdims=True))             # weight_norm(dim=2)
    put(np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(w.shape[0], -1))      # [H][tap*cg + ci] = [G][cg][...]
    put(g("encoder.pos_conv_embed.conv.bias"))
    put(g("encoder.layer_norm.weight")); put(g("encoder.layer_norm.bias"))
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}."
        put(np.concatenate([g(p + f"attention.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], axis=0))
        put(np.concatenate([g(p + f"attention.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")]))
        put(g(p + "attention.out_proj.weight")); put(g(p + "attention.out_proj.bias"))
        put(g(p + "layer_norm.weight")); put(g(p + "layer_norm.bias"))
        put(g(p + "feed_forward.intermediate_dense.weight")); put(g(p + "feed_forward.intermediate_dense.bias"))
        put(g(p + "feed_forward.output_dense.weight")); put(g(p + "feed_forward.output_dense.bias"))
        put(g(p + "final_layer_norm.weight")); put(g(p + "final_layer_norm.bias"))
    return blob


class W2V2Engine:
    """Device-resident Wav2Vec2 weights + workspace; runs batches of windows through the HIP path."""

    def __init__(self, cfg: W2V2Config, sd: dict, device="cuda", max_chunks_per_call: int = 256):
        import torch
        _lib.load()
        _lib.require_gpu()
        self.cfg = cfg
        self.device = torch.device(device)
        self.blob = torch.from_numpy(pack_weights(cfg, sd)).to(self.device)
        lim = 65535 // max(cfg.num_attention_heads, cfg.num_conv_pos_embedding_groups)
        self.max_chunks = max(1, min(max_chunks_per_call, lim))
        self._ws = None
        self._keep = None

    def _workspace(self, lens_c, n):
        import torch
        need = _lib.load().rsaf_w2v2_workspace_bytes_ragged(lens_c, n, *_cfg_args(self.cfg))
        if need < 0:
            raise _lib.RsafError("a window is shorter than the encoder's receptive field (or the lengths are not non-increasing)")
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = None
            self._ws = torch.empty(need // 4 + 4, dtype=torch.float32, device=self.device)
        return self._ws

    def forward_windows(self, wav, starts, lens, out, out_rows, stream=None):
        """wav: 1-D float32 device tensor; starts / lens / out_rows: host arrays (sample offset and length of each window in
        ``wav``; first output row of each window in ``out`` [rows, hidden]).  Windows of any mix of lengths run together
        (``rsaf_w2v2_forward_ragged``): they are ordered by length here, longest first, and cut into balanced sub-batches."""
        import torch
        lib = _lib.load()
        cfg = self.cfg
        starts = np.asarray(starts, dtype=np.int64)
        lens = np.asarray(lens, dtype=np.int32)
        out_rows = np.asarray(out_rows, dtype=np.int64)
        n_total = len(starts)
        if n_total == 0:
            return out
        order = np.argsort(-lens.astype(np.int64), kind="stable")
        starts, lens, out_rows = starts[order], lens[order], out_rows[order]
        # balanced sub-batches: ceil(n / max) calls of (almost) equal size instead of full ones and a remainder (7 000 windows
        # with a maximum of 2 048: 4 x 1 750, whose GEMM tile counts fill the 256 CUs to 99.8 % where 2 048 left the last round
        # of every N = 768 GEMM a third full)
        n_calls = max(1, -(-n_total // self.max_chunks))
        per = max(1, -(-n_total // n_calls))
        per = min(self.max_chunks, (per + 3) & ~3)
        # one pinned staging buffer per call for the three small tables, queued on the current stream (no host wait)
        for b0 in range(0, n_total, per):
            n = min(per, n_total - b0)
            host = torch.empty(5 * n, dtype=torch.int32).pin_memory()          # starts (int64) | out rows (int64) | lengths (int32)
            hv = host.numpy()
            hv[:2 * n].view(np.int64)[:] = starts[b0:b0 + n]
            hv[2 * n:4 * n].view(np.int64)[:] = out_rows[b0:b0 + n]
            hv[4 * n:] = lens[b0:b0 + n]
            dev = host.to(self.device, non_blocking=True)
            lens_c = (C.c_int * n)(*[int(v) for v in lens[b0:b0 + n]])
            ws = self._workspace(lens_c, n)
            _lib.check(lib.rsaf_w2v2_forward_ragged(
                _lib.ptr(wav), C.c_void_p(dev.data_ptr()), C.c_void_p(dev.data_ptr() + 16 * n), lens_c, n,
                *_cfg_args(cfg), float(cfg.layer_norm_eps), _lib.ptr(self.blob), _lib.ptr(ws), ws.numel() * 4, _lib.ptr(out),
                C.c_void_p(dev.data_ptr() + 8 * n), _lib.stream_ptr(stream)), "rsaf_w2v2_forward_ragged")
            self._keep = (host, dev)                                           # alive until the next call's copy is queued
        return out

    def plan(self, lengths, chunk_seconds=5, overlap_seconds=1):
        """Integer-exact window plan of a batch: per clip [(start, len, frames)] + total frames."""
        per_clip, totals = [], []
        for n in lengths:
            pl = [(s, l, self.cfg.frames(l)) for s, l in chunk_plan(n, chunk_seconds, overlap_seconds)]
            per_clip.append(pl)
            totals.append(sum(f for _, _, f in pl))
        return per_clip, totals

    def extract_packed(self, wav, clip_offsets, lengths, chunk_seconds=5, overlap_seconds=1, stream=None):
        """All clips of a packed batch -> (out [sum frames, hidden] device tensor, frame offsets).

        wav: 1-D float32 device tensor holding the clips back to back; clip c = samples
        [clip_offsets[c], clip_offsets[c] + lengths[c])."""
        import torch
        per_clip, totals = self.plan(lengths, chunk_seconds, overlap_seconds)
        frame_off = np.zeros(len(lengths) + 1, dtype=np.int64)
        frame_off[1:] = np.cumsum(totals)
        out = torch.empty((max(int(frame_off[-1]), 1), self.cfg.hidden_size), dtype=torch.float32, device=self.device)
        starts, lens, rows = [], [], []
        for c, pl in enumerate(per_clip):
            row = int(frame_off[c])
            for s, l, f in pl:
                starts.append(int(clip_offsets[c]) + s)
                lens.append(l)
                rows.append(row)
                row += f
        self.forward_windows(wav, starts, lens, out, rows, stream)
        return out[:int(frame_off[-1])], frame_off


_ENGINES = {}


def _resolve_model(model_name):
    """(cfg, state_dict): a local HF directory, or seeded random base weights when the caller sets
    RSAF_W2V2_RANDOM_SEED (benchmarks / parity tests).  Never fetches."""
    seed = os.environ.get("RSAF_W2V2_RANDOM_SEED")
    if os.path.isdir(str(model_name)):
        return load_local_model(str(model_name))
    if seed is not None:
        cfg = W2V2Config()
        return cfg, random_state_dict(cfg, int(seed))
    raise FileNotFoundError(
        f"'{model_name}' is not a local model directory and this build never downloads checkpoints "
        "(no network): pass a directory holding config.json + model.safetensors")


def get_engine(model_name, device="cuda"):
    key = (str(model_name), os.environ.get("RSAF_W2V2_RANDOM_SEED"), str(device))
    if key not in _ENGINES:
        cfg, sd = _resolve_model(model_name)
        _ENGINES[key] = W2V2Engine(cfg, sd, device)
    return _ENGINES[key]


def extract_wav2vec2_sequences(input_df, model_name="facebook/wav2vec2-base-960h", audio_file_column="filepath",
                               chunk_seconds=5, overlap_seconds=1, verbose=True, batch_files=64):
    """Drop-in for ``src/foundation_model_extractor.py:37-131``: dict basename -> float32 [T, 768].

    Files shorter than 0.5 s (``:88``) or failing to load are absent; a model that cannot be
    loaded gives the reference's convention ``print + {}`` (``:73-74``)."""
    import torch
    device = "cuda" if torch.cuda.is_available() else "cpu"
    if verbose:
        print(f"Using device: {device}")
    try:
        _lib.load()
        _lib.require_gpu()
        eng = get_engine(model_name, device)
    except Exception as e:
        print(f"Error loading model '{model_name}': {e}")
        return {}
    sequences = {}
    paths = list(input_df[audio_file_column])
    for b0 in range(0, len(paths), batch_files):
        clips, names = [], []
        for pth in paths[b0:b0 + batch_files]:
            filename = os.path.basename(pth)
            try:
                mono, fs, n_in = read_wav_mono_device(pth, device=eng.device)   # :87,91 decode + channel mean on the device
                if n_in < int(SAMPLE_RATE * 0.5):                             # :88 (pre-resample count)
                    if verbose:
                        print(f"INFO: Skipping very short file '{filename}'.")
                    continue
                if fs != SAMPLE_RATE:                                         # :92-94 torchaudio Resample defaults
                    from .resample import resample_sinc_hann
                    mono = resample_sinc_hann(mono, fs, SAMPLE_RATE, device=eng.device)
                clips.append(mono)
                names.append(filename)
            except Exception as e:
                if verbose:
                    print(f"FATAL ERROR processing file '{filename}': {e}. Skipping.")
        if not clips:
            continue
        def run(batch_clips):
            lengths = [int(c.numel()) for c in batch_clips]
            offs = np.zeros(len(batch_clips) + 1, dtype=np.int64)
            offs[1:] = np.cumsum(lengths)
            wav = torch.cat(batch_clips) if len(batch_clips) > 1 else batch_clips[0].contiguous()
            out, frame_off = eng.extract_packed(wav, offs[:-1], lengths, chunk_seconds, overlap_seconds)
            torch.cuda.synchronize()
            return out.cpu().numpy(), frame_off

        try:
            host, frame_off = run(clips)
            for i, fn in enumerate(names):
                a, b = int(frame_off[i]), int(frame_off[i + 1])
                if b > a:                                                      # :123 (no chunk survived)
                    sequences[fn] = host[a:b].copy()
        except (_lib.RsafError, torch.cuda.OutOfMemoryError) as e:
            # one bad file (or an allocation failure of the whole batch) must not take the other files of the batch or
            # the files already done with it: the reference skips only the offending file (:127-129).  Any other
            # RuntimeError (a HIP fault surfacing at the synchronize) is NOT retried on the poisoned context: it propagates.
            if verbose:
                print(f"WARNING: batch of {len(clips)} files failed ({e}); retrying file by file.")
            for fn, c in zip(names, clips):
                try:
                    host, frame_off = run([c])
                    if int(frame_off[1]) > 0:
                        sequences[fn] = host[:int(frame_off[1])].copy()
                except (_lib.RsafError, torch.cuda.OutOfMemoryError) as e1:
                    if verbose:
                        print(f"FATAL ERROR processing file '{fn}': {e1}. Skipping.")
    return sequences


def extract_wav2vec2_embeddings(input_df, **kwargs):
    """Drop-in for ``src/foundation_model_extractor.py:133-166``: time-mean per file."""
    import pandas as pd
    seqs = extract_wav2vec2_sequences(input_df, **kwargs)
    if not seqs:
        return pd.DataFrame()
    rows = []
    for filename, seq in seqs.items():
        m = np.mean(seq, axis=0)
        d = {f"dim_{k}": v for k, v in enumerate(m)}
        d["filename"] = filename
        rows.append(d)
    return pd.DataFrame(rows)
