"""HIP Wav2Vec2 path vs transformers goldens (small geometry) and vs the oracle (base geometry)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import w2v2_oracle as wo
from robust_speech_analysis_framework_amd import synth
from robust_speech_analysis_framework_amd.w2v2_config import W2V2Config, random_state_dict, save_local_model

TOL = 1e-4      # north_star: <= 1e-4 relative for float outputs


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _small():
    z = np.load(os.path.join(HERE, "golden", "w2v2_small.npz"))
    cfg = W2V2Config(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(z["cfg"])).items()})
    return z, cfg, random_state_dict(cfg, seed=7)


def _windows(eng, wav_np, starts, length):
    import torch
    T = eng.cfg.frames(length)
    wav = torch.from_numpy(np.ascontiguousarray(wav_np, dtype=np.float32)).cuda()
    out = torch.empty((len(starts) * T, eng.cfg.hidden_size), dtype=torch.float32, device="cuda")
    eng.forward_windows(wav, np.asarray(starts), [length] * len(starts), out, np.arange(len(starts)) * T)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(len(starts), T, -1)


@pytest.mark.parametrize("n", [8000, 20000])
def test_small_geometry_matches_transformers_golden(rsaf_lib, n):
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    z, cfg, sd = _small()
    eng = W2V2Engine(cfg, sd)
    clip = synth.synth_clip(50, 2.0)[:n]
    got = _windows(eng, clip, [0], n)[0]
    assert got.shape == z[f"last_hidden_state_{n}"].shape
    assert _rel(got, z[f"last_hidden_state_{n}"]) < TOL


def test_ragged_clips_follow_reference_chunk_loop(rsaf_lib):
    """Windows of 80 000 / every 64 000, per-window normalisation, vstack with duplicated overlap."""
    import torch
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    z, cfg, sd = _small()
    eng = W2V2Engine(cfg, sd, max_chunks_per_call=3)          # force several sub-batches
    secs = [5.5, 0.6, 11.0, 4.5, 0.4999, 5.0]
    clips = [synth.synth_clip(60 + i, s) for i, s in enumerate(secs)]
    lengths = [len(c) for c in clips]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    wav = torch.from_numpy(np.concatenate(clips)).cuda()
    out, frame_off = eng.extract_packed(wav, offs[:-1], lengths)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for i, c in enumerate(clips):
        ref = wo.extract_sequence(sd, cfg, c)
        a, b = int(frame_off[i]), int(frame_off[i + 1])
        if ref is None:
            assert a == b
            continue
        assert (b - a, cfg.hidden_size) == ref.shape                    # integer-exact frame counts
        assert _rel(host[a:b], ref) < TOL
    assert int(frame_off[6] - frame_off[5]) == cfg.frames(80000) + cfg.frames(16000)   # 5 s = TWO windows


def test_base_geometry_window_matches_oracle(rsaf_lib):
    """wav2vec2-base geometry (seeded random weights): one 5 s and one 2 s window, batch of 3."""
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    cfg = W2V2Config()
    sd = random_state_dict(cfg, seed=0)
    eng = W2V2Engine(cfg, sd)
    clip = synth.synth_clip(70, 9.0)
    got = _windows(eng, clip, [0, 64000], 80000)
    assert got.shape == (2, 249, 768)
    for j, s in enumerate((0, 64000)):
        ref = wo.forward(sd, cfg, wo.hf_normalize(clip[s:s + 80000])[None])[0]
        assert _rel(got[j], ref) < TOL, (j, _rel(got[j], ref))
    tail = _windows(eng, clip, [100], 32000)
    assert tail.shape == (1, 99, 768)
    assert _rel(tail[0], wo.forward(sd, cfg, wo.hf_normalize(clip[100:32100])[None])[0]) < TOL


def test_group_width_not_multiple_of_16_matches_oracle(rsaf_lib):
    """The positional convolution runs on the f16x3 GEMM when its group width is a multiple of 16 (base: 48, the small
    golden geometry: 16) and on the fp32 MFMA GEMM otherwise: a geometry with 8-wide groups (and 16-wide heads: the
    unfused attention with its panel split) against the oracle."""
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    cfg = W2V2Config(conv_dim=(32,) * 7, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                     num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=8)
    sd = random_state_dict(cfg, seed=3)
    eng = W2V2Engine(cfg, sd)
    clip = synth.synth_clip(71, 3.0)
    got = _windows(eng, clip, [0, 8000], 32000)
    for j, s in enumerate((0, 8000)):
        ref = wo.forward(sd, cfg, wo.hf_normalize(clip[s:s + 32000])[None])[0]
        assert got[j].shape == ref.shape
        assert _rel(got[j], ref) < TOL, (j, _rel(got[j], ref))


def test_30s_clip_frame_count_and_batch_independence(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    cfg = W2V2Config()
    eng = W2V2Engine(cfg, random_state_dict(cfg, seed=0))
    clips = [synth.synth_clip(80 + i, 30.0) for i in range(2)]
    wav = torch.from_numpy(np.concatenate(clips)).cuda()
    out, fo = eng.extract_packed(wav, [0, 480000], [480000, 480000])
    torch.cuda.synchronize()
    assert list(fo) == [0, 1842, 3684]
    alone, _ = eng.extract_packed(wav[480000:].contiguous(), [0], [480000])
    torch.cuda.synchronize()
    # a window's values do not depend on what else is in the batch (the reference runs batch 1)
    assert torch.equal(out[1842:], alone)


def test_dropin_sequences_and_embeddings(rsaf_lib, tmp_path, monkeypatch):
    import pandas as pd
    from robust_speech_analysis_framework_amd import w2v2
    z, cfg, sd = _small()
    mdir = tmp_path / "model"
    save_local_model(str(mdir), cfg, sd)
    paths = synth.write_synth_corpus(str(tmp_path / "wav"), 2, 5.5, first=60)
    short = tmp_path / "wav" / "short.wav"
    synth.write_wav(str(short), synth.synth_clip_int16(99, 0.3))
    bad = tmp_path / "wav" / "bad.wav"
    bad.write_bytes(b"RIFFxxxx")
    df = pd.DataFrame({"filepath": [paths[0], str(short), str(bad), paths[1]]})
    seqs = w2v2.extract_wav2vec2_sequences(df, model_name=str(mdir), verbose=False)
    assert list(seqs) == ["synth_00060.wav", "synth_00061.wav"]
    ref = wo.extract_sequence(sd, cfg, synth.synth_clip(60, 5.5))
    assert seqs["synth_00060.wav"].dtype == np.float32 and _rel(seqs["synth_00060.wav"], ref) < TOL
    emb = w2v2.extract_wav2vec2_embeddings(df, model_name=str(mdir), verbose=False)
    assert list(emb.columns) == [f"dim_{k}" for k in range(cfg.hidden_size)] + ["filename"]
    assert np.allclose(emb.iloc[0, :-1].to_numpy(dtype=np.float64), ref.mean(axis=0), atol=1e-4)
    # model that is not a local directory: reference convention print + {} (no fetch is attempted)
    monkeypatch.delenv("RSAF_W2V2_RANDOM_SEED", raising=False)
    assert w2v2.extract_wav2vec2_sequences(df, model_name="facebook/wav2vec2-base-960h", verbose=False) == {}
    assert w2v2.extract_wav2vec2_embeddings(df, model_name="facebook/wav2vec2-base-960h", verbose=False).empty


def test_twelve_ragged_clips_in_one_call_match_the_oracle_at_base_geometry(rsaf_lib):
    """The reference ends every file in a tail window of its own length (src/foundation_model_extractor.py:103-108): twelve
    clips of 0.6 - 11 s give 12 distinct tail lengths beside the full windows.  All of them run in ONE
    rsaf_w2v2_forward_ragged call (per-window normalisation, GroupNorm statistics, zero padding of the positional
    convolution and attention over each window's own frames) and match the per-window torch-CPU restatement."""
    import torch
    from robust_speech_analysis_framework_amd import _lib
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    cfg = W2V2Config()
    sd = random_state_dict(cfg, seed=0)
    eng = W2V2Engine(cfg, sd, max_chunks_per_call=64)
    secs = [0.6, 1.3, 2.05, 3.3, 4.4, 5.0, 5.35, 6.7, 7.9, 9.0, 10.1, 11.0]
    clips = [synth.synth_clip(500 + i, s) for i, s in enumerate(secs)]
    lengths = [len(c) for c in clips]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    wav = torch.from_numpy(np.concatenate(clips)).cuda()
    calls = []
    lib = _lib.load()
    real = lib.rsaf_w2v2_forward_ragged

    class Spy:                                                  # counts the library calls and the windows each one took
        def __call__(self, *a):
            calls.append(int(a[4]))
            return real(*a)
    lib.rsaf_w2v2_forward_ragged = Spy()
    try:
        out, frame_off = eng.extract_packed(wav, offs[:-1], lengths)
        torch.cuda.synchronize()
    finally:
        lib.rsaf_w2v2_forward_ragged = real
    n_windows = sum(len(eng.plan([n])[0][0]) for n in lengths)
    assert calls == [n_windows] and n_windows >= 20           # every window of every clip in one call
    host = out.cpu().numpy()
    worst = 0.0
    for i, c in enumerate(clips):
        ref = wo.extract_sequence(sd, cfg, c)
        a, b = int(frame_off[i]), int(frame_off[i + 1])
        assert (b - a, cfg.hidden_size) == ref.shape          # integer-exact frame counts
        worst = max(worst, _rel(host[a:b], ref))
        rows = np.abs(host[a:b] - ref).max(axis=1) / np.abs(ref).max(axis=1)
        assert rows.max() < TOL, (i, int(np.argmax(rows)))
    assert worst < TOL


def test_ragged_call_returns_the_bits_of_the_per_length_calls(rsaf_lib):
    """A window's frames must not depend on its batch mates: the mixed-length call and one call per window give the same bits."""
    import torch
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    z, cfg, sd = _small()
    eng = W2V2Engine(cfg, sd)
    clip = synth.synth_clip(77, 7.0)
    spec = [(0, 80000), (1000, 52000), (64000, 48000), (30000, 9000), (5, 80000), (200, 400)]
    T = [cfg.frames(l) for _, l in spec]
    rows = np.concatenate([[0], np.cumsum(T)])
    wav = torch.from_numpy(clip).cuda()
    together = torch.zeros((int(rows[-1]), cfg.hidden_size), device="cuda")
    eng.forward_windows(wav, [s for s, _ in spec], [l for _, l in spec], together, rows[:-1])
    alone = torch.zeros_like(together)
    for k, (s0, l) in enumerate(spec):
        eng.forward_windows(wav, [s0], [l], alone, [int(rows[k])])
    torch.cuda.synchronize()
    assert torch.equal(together, alone)


def test_positional_conv_kernel_matches_the_gemm_form_and_the_oracle_on_10s_windows(rsaf_lib, monkeypatch):
    """The positional convolution runs as its own kernel with the window's panel image resident in LDS
    (csrc/w2v2.hip: posconv_f16x3_kernel; split K for windows of up to 256 frames, 512 rows per workgroup above).  Both
    variants against the batched-GEMM form it replaced (RSAF_W2V2_POSCONV_GEMM=1: the same three-product arithmetic, another
    summation order) on 5 s windows (249 frames), 10 s windows (499 frames) and a ragged mix, and the 10 s window - the
    variant no other test reaches - against the numpy oracle."""
    import torch
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    cfg = W2V2Config()
    sd = random_state_dict(cfg, seed=11)
    eng = W2V2Engine(cfg, sd)
    clip = synth.synth_clip(402, 21.0)
    wav = torch.from_numpy(clip).cuda()
    cases = [[(0, 80000), (64000, 80000), (128000, 80000)],
             [(0, 160000), (100000, 160000)],
             [(0, 160000), (3000, 100000), (50000, 80000), (7, 30000), (90000, 2000)]]
    for spec in cases:
        T = [cfg.frames(l) for _, l in spec]
        rows = np.concatenate([[0], np.cumsum(T)])
        outs = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("RSAF_W2V2_POSCONV_GEMM", mode)
            out = torch.full((int(rows[-1]), cfg.hidden_size), float("nan"), dtype=torch.float32, device="cuda")
            eng.forward_windows(wav, [s for s, _ in spec], [l for _, l in spec], out, rows[:-1])
            torch.cuda.synchronize()
            outs[mode] = out.cpu().numpy()
        monkeypatch.delenv("RSAF_W2V2_POSCONV_GEMM", raising=False)
        assert np.isfinite(outs["0"]).all()
        scale = np.abs(outs["1"]).max()
        print("positional conv A/B", [l for _, l in spec], np.abs(outs["0"] - outs["1"]).max() / scale)
        assert np.abs(outs["0"] - outs["1"]).max() <= 1e-5 * scale, (spec, np.abs(outs["0"] - outs["1"]).max(), scale)
    ref = wo.forward(sd, cfg, wo.hf_normalize(clip[:160000])[None])[0]
    got = outs["0"][:cfg.frames(160000)]
    assert got.shape == ref.shape == (499, cfg.hidden_size)
    assert _rel(got, ref) < TOL, _rel(got, ref)
