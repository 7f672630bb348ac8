"""SURVEY.md §8f rank 4 on REAL drop-in output: what the notebooks do with the extractors' results (to_csv / read_csv of the
feature tables merged with metadata, pickle of the sequence dict, torch.save of a checkpoint dict) round-trips, and the
re-read data drive the next stage (session aggregation, CNN-LSTM forward) to the same numbers."""
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_extractor_outputs_round_trip_through_the_notebooks_cache_formats(rsaf_lib, tmp_path, monkeypatch):
    import pandas as pd
    import torch
    from robust_speech_analysis_framework_amd import synth
    from src.foundation_model_extractor import extract_wav2vec2_embeddings, extract_wav2vec2_sequences
    from src.models import CNNLSTM
    from src.mshds_extractor import extract_mshds_features
    from src.opensmile_extractor import extract_opensmile_features
    from src.utils import aggregate_clip_features, aggregate_interview_sequences
    from tests.test_smile_gpu import _MINI_CONF

    monkeypatch.setenv("RSAF_W2V2_RANDOM_SEED", "0")                   # seeded base-geometry weights (no checkpoint offline)
    paths = synth.write_synth_corpus(str(tmp_path), 4, 1.2, first=300)
    meta = pd.DataFrame({"filepath": paths, "filename": [p.split("/")[-1] for p in paths],
                         "unique_participant_id": ["P1", "P1", "P2", "P2"], "label": ["Control", "Control", "Patient", "Patient"]})
    conf = tmp_path / "Androids.conf"
    conf.write_text(_MINI_CONF)

    # notebooks/01: feature tables -> merge with metadata -> to_csv(index=False) -> read_csv
    for name, df in (("mshds", extract_mshds_features(meta, verbose=False)),
                     ("opensmile", extract_opensmile_features(meta, "unused", str(conf), verbose=False)),
                     ("wav2vec2", extract_wav2vec2_embeddings(meta, verbose=False))):
        assert len(df) == 4 and "filename" in df.columns, name
        merged = pd.merge(meta, df, on="filename", how="left")
        out = tmp_path / f"features_{name}.csv"
        merged.to_csv(out, index=False)
        back = pd.read_csv(out)
        assert list(back.columns) == list(merged.columns)
        num = [c for c in df.columns if c != "filename"]
        a, b = merged[num].to_numpy(dtype=np.float64), back[num].to_numpy(dtype=np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        if df[num[0]].dtype == np.float32:                             # float32 columns are written with their shortest repr:
            assert np.array_equal(merged[num].to_numpy(dtype=np.float32), back[num].to_numpy(dtype=np.float32)), name   # exact as float32
        else:
            assert np.allclose(a, b, rtol=1e-12, atol=0, equal_nan=True), name
        # session aggregation on the re-read table gives what it gives on the live one
        agg_live = aggregate_clip_features(df, meta)
        agg_back = aggregate_clip_features(back[["filename"] + num], meta)
        assert list(agg_live.columns) == list(agg_back.columns) and len(agg_back) == 2
        va, vb = agg_live.iloc[:, 1:].to_numpy(dtype=np.float64), agg_back.iloc[:, 1:].to_numpy(dtype=np.float64)
        assert np.allclose(va, vb, rtol=1e-5, atol=1e-6, equal_nan=True), name   # float32 columns come back within their own rounding

    # notebooks/03: dict[str, float32 ndarray] -> pickle -> aggregate per participant -> CNN-LSTM
    seqs = extract_wav2vec2_sequences(meta, verbose=False)
    assert set(seqs) == set(meta["filename"]) and all(v.dtype == np.float32 and v.shape[1] == 768 for v in seqs.values())
    pk = tmp_path / "wav2vec2_sequences.pkl"
    with open(pk, "wb") as f:
        pickle.dump(seqs, f)
    with open(pk, "rb") as f:
        seqs_back = pickle.load(f)
    assert all(np.array_equal(seqs[k], seqs_back[k]) for k in seqs)
    per_part = aggregate_interview_sequences(seqs_back, meta)
    assert set(per_part) == {"P1", "P2"} and per_part["P1"].shape[0] == seqs["synth_00300.wav"].shape[0] + seqs["synth_00301.wav"].shape[0]

    torch.manual_seed(0)
    model = CNNLSTM(cnn_out_channels=32, lstm_hidden_dim=64).cuda().eval()
    x = torch.from_numpy(per_part["P1"][None]).cuda()
    want = model(x).cpu()
    ck = tmp_path / "final_tuned_cnn_lstm.pt"
    torch.save({"model_state_dict": model.state_dict(), "hyperparameters": {"cnn_out_channels": 32, "lstm_hidden_dim": 64}}, ck)
    saved = torch.load(ck, map_location="cpu", weights_only=True)
    m2 = CNNLSTM(cnn_out_channels=32, lstm_hidden_dim=64)
    m2.load_state_dict(saved["model_state_dict"])
    assert torch.equal(m2.cuda().eval()(x).cpu(), want)                # reloaded checkpoint -> bit-identical logits
