"""CPU restatement of ``src/utils.py`` (session aggregation).  TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

PINNED: the reference module itself imports and runs in the build container (pandas / numpy / tqdm only); the
fixtures ``tests/golden/aggregate_*.json|npz`` hold its outputs on seeded inputs
(``tests/golden/make_aggregate_golden.py``).  This file restates the arithmetic without pandas' group-by:
``aggregate_clip_features`` = inner join on ``filename`` in metadata order, groups sorted by participant id,
per column the mean and the sample standard deviation (n - 1) of the non-NaN values (NaN below 1 / 2 values)
(``src/utils.py:39-59``); ``aggregate_interview_sequences`` = stack the available clip sequences of each
participant in metadata order (``src/utils.py:83-102``).
"""
from __future__ import annotations

import numpy as np


def aggregate_clip_features(filenames, values, meta_filenames, meta_participants):
    """filenames [n], values float64 [n, width]; metadata rows (filename, participant).
    Returns (participants sorted, out float64 [n_part, width, 2])."""
    values = np.asarray(values, dtype=np.float64)
    where = {}
    for i, f in enumerate(filenames):
        where.setdefault(f, []).append(i)
    groups = {}
    for f, p in zip(meta_filenames, meta_participants):
        if p is None or (isinstance(p, float) and np.isnan(p)):
            continue
        for i in where.get(f, []):
            groups.setdefault(p, []).append(i)
    keys = sorted(groups)
    out = np.full((len(keys), values.shape[1], 2), np.nan)
    for g, k in enumerate(keys):
        blk = values[groups[k]]
        for c in range(values.shape[1]):
            v = blk[:, c]
            v = v[~np.isnan(v)]
            if len(v) >= 1:
                m = float(np.sum(v)) / len(v)
                out[g, c, 0] = m
                if len(v) >= 2:
                    out[g, c, 1] = np.sqrt(float(np.sum((v - m) ** 2)) / (len(v) - 1))
    return keys, out


def aggregate_interview_sequences(clip_sequences, meta_filenames, meta_participants):
    groups = {}
    for f, p in zip(meta_filenames, meta_participants):
        groups.setdefault(p, []).append(f)
    out = {}
    for p in sorted(groups):
        seqs = [np.asarray(clip_sequences[f]) for f in groups[p] if f in clip_sequences]
        if seqs:
            out[p] = np.concatenate(seqs, axis=0)
    return out
