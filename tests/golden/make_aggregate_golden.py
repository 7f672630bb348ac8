"""Generate tests/golden/aggregate_golden.npz from the REFERENCE module (src/utils.py).

Run in the build container only (needs /root/reference), from any directory:
    python tests/golden/make_aggregate_golden.py
The reference never travels; what is committed is data: seeded inputs and the reference's outputs.
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, "/root/reference")
from src.utils import aggregate_clip_features, aggregate_interview_sequences  # noqa: E402  (the reference)


def make_inputs(seed=20260701, n_clips=57, n_part=9, width=13):
    rng = np.random.Generator(np.random.PCG64(seed))
    files = [f"clip_{i:03d}.wav" for i in range(n_clips)]
    parts = [f"P{int(rng.integers(0, n_part)):02d}_s{int(rng.integers(1, 3))}" for _ in range(n_clips)]
    vals = rng.standard_normal((n_clips, width)) * rng.uniform(0.1, 100.0, size=width) + rng.uniform(-5, 5, size=width)
    vals[rng.random(vals.shape) < 0.08] = np.nan                      # failed helpers give NaN features
    vals[:, 3] = np.where(np.array(parts) == parts[0], np.nan, vals[:, 3])   # one participant: a column all NaN
    meta_order = rng.permutation(n_clips)
    meta_files = [files[i] for i in meta_order] + ["missing_a.wav", "missing_b.wav"]      # clips that failed extraction
    meta_parts = [parts[i] for i in meta_order] + [parts[1], "P99_s1"]
    keep = rng.random(n_clips) < 0.9                                   # feature rows exist for 90 % of the clips
    keep[:3] = True
    seq_len = rng.integers(1, 9, size=n_clips)
    return files, parts, vals, meta_files, meta_parts, keep, seq_len


def main():
    files, parts, vals, meta_files, meta_parts, keep, seq_len = make_inputs()
    cols = [f"feat_{j}" for j in range(vals.shape[1])]
    feat_df = pd.DataFrame(vals[keep], columns=cols)
    feat_df.insert(0, "filename", [f for f, k in zip(files, keep) if k])
    meta_df = pd.DataFrame({"filename": meta_files, "unique_participant_id": meta_parts, "other": 1})
    ref = aggregate_clip_features(feat_df, meta_df)
    seqs = {f: (np.arange(n * 5, dtype=np.float32).reshape(n, 5) + 1000.0 * i)
            for i, (f, n, k) in enumerate(zip(files, seq_len, keep)) if k}
    ref_seq = aggregate_interview_sequences(seqs, meta_df)
    out = {
        "files": np.array(files), "keep": keep, "values": vals, "meta_files": np.array(meta_files),
        "meta_parts": np.array(meta_parts), "seq_len": seq_len,
        "ref_columns": np.array(list(ref.columns)), "ref_participants": ref["unique_participant_id"].to_numpy().astype(str),
        "ref_values": ref.drop(columns=["unique_participant_id"]).to_numpy(dtype=np.float64),
        "ref_seq_keys": np.array(sorted(ref_seq)),
    }
    for k in ref_seq:
        out[f"ref_seq__{k}"] = ref_seq[k]
    np.savez_compressed(os.path.join(HERE, "aggregate_golden.npz"), **out)
    print("participants", len(ref), "columns", len(ref.columns), "sessions with sequences", len(ref_seq))


if __name__ == "__main__":
    main()
