"""The CPU restatement of src/utils.py against the reference's own outputs (golden fixture, no GPU)."""
import os

import numpy as np

from oracle import aggregate_oracle as ao

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "aggregate_golden.npz"), allow_pickle=False)


def _inputs():
    files = [str(f) for f in G["files"]]
    keep = G["keep"]
    kept = [f for f, k in zip(files, keep) if k]
    return files, keep, kept, G["values"][keep], [str(f) for f in G["meta_files"]], [str(p) for p in G["meta_parts"]]


def test_clip_features_match_reference_groupby():
    files, keep, kept, vals, mf, mp = _inputs()
    keys, out = ao.aggregate_clip_features(kept, vals, mf, mp)
    assert keys == [str(p) for p in G["ref_participants"]]                      # sorted groups, absent participants dropped
    ref = G["ref_values"].reshape(len(keys), -1, 2)
    assert np.array_equal(np.isnan(out), np.isnan(ref))                          # NaN pattern (all-NaN column, single value)
    ok = ~np.isnan(ref)
    assert np.abs(out[ok] - ref[ok]).max() <= 1e-12 * np.abs(ref[ok]).max()
    cols = [str(c) for c in G["ref_columns"]]
    assert cols[0] == "unique_participant_id" and cols[1:3] == ["feat_0_mean", "feat_0_std"]


def test_interview_sequences_match_reference_stacking():
    files, keep, kept, vals, mf, mp = _inputs()
    seqs = {f: (np.arange(n * 5, dtype=np.float32).reshape(n, 5) + 1000.0 * i)
            for i, (f, n, k) in enumerate(zip(files, G["seq_len"], keep)) if k}
    out = ao.aggregate_interview_sequences(seqs, mf, mp)
    assert sorted(out) == [str(k) for k in G["ref_seq_keys"]]
    for k in out:
        assert np.array_equal(out[k], G[f"ref_seq__{k}"])                        # exact: a stack of copies
