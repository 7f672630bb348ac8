"""Session aggregation and batch assembly behind the extractors.

Drop-ins for ``src/utils.py``: ``aggregate_clip_features`` (:7-61) and ``aggregate_interview_sequences``
(:64-104), same signatures, column names, ordering and skip rules; the relational part (the merge on
``filename``, the sorted participant groups) stays in pandas on the host, the arithmetic (per-participant
mean / sample standard deviation of every feature column, the row stacking) runs on the device through
``librsaf.so``.  ``pad_batch_device`` is the device form of ``collate_fn`` (``src/dl_cv_strategies.py:81-84``)
for sequences that are already resident in HBM.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def segment_mean_std(rows, row_index, seg_off, stream=None):
    """rows: float64 [n, width] (torch, device); row_index int32 [m]; seg_off int32 [n_seg + 1] ->
    float64 [n_seg, width, 2] = (mean, std ddof=1) with NaN skipped."""
    import torch
    lib = _lib.load()
    _lib.require_gpu()
    n_seg, width = int(len(seg_off)) - 1, int(rows.shape[1])
    out = torch.empty((max(n_seg, 0), width, 2), dtype=torch.float64, device=rows.device)
    if n_seg > 0 and width > 0:
        ri = torch.as_tensor(np.asarray(row_index, dtype=np.int32), device=rows.device)
        so = torch.as_tensor(np.asarray(seg_off, dtype=np.int32), device=rows.device)
        _lib.check(lib.rsaf_segment_mean_std(_lib.ptr(rows), int(rows.stride(0)), _lib.ptr(ri), _lib.ptr(so), n_seg, width,
                                             _lib.ptr(out), _lib.stream_ptr(stream)), "rsaf_segment_mean_std")
    return out


def gather_rows(src, src_row, out=None, stream=None):
    """src float32 [n, width] (device); src_row int64 [m] (host or device, -1 = zero row) -> float32 [m, width]."""
    import torch
    lib = _lib.load()
    _lib.require_gpu()
    idx = torch.as_tensor(np.asarray(src_row, dtype=np.int64) if not torch.is_tensor(src_row) else src_row,
                          dtype=torch.int64, device=src.device).contiguous()
    m, width = int(idx.numel()), int(src.shape[1])
    if out is None:
        out = torch.empty((m, width), dtype=torch.float32, device=src.device)
    if m and width:
        _lib.check(lib.rsaf_gather_rows_f32(_lib.ptr(src), int(src.stride(0)), _lib.ptr(idx), m, width, _lib.ptr(out),
                                            int(out.stride(0)), _lib.stream_ptr(stream)), "rsaf_gather_rows_f32")
    return out


def aggregate_clip_features(clip_features_df, metadata_df):
    """Drop-in for ``src/utils.py:7-61``: one row per participant (sorted), columns
    ``unique_participant_id`` then ``<feature>_mean``, ``<feature>_std`` per feature column."""
    import pandas as pd
    import torch
    if clip_features_df.empty:
        print("Warning: Input clip_features_df is empty. Return an empty aggregated DataFrame.")
        return pd.DataFrame()
    metadata_subset = metadata_df[["filename", "unique_participant_id"]]
    merged = pd.merge(metadata_subset, clip_features_df, on="filename").drop(columns=["filename"])     # :39-42
    feature_cols = [c for c in merged.columns if c != "unique_participant_id"]
    codes, uniques = pd.factorize(merged["unique_participant_id"], sort=True)                           # groupby sorts, drops NaN keys
    keep = np.flatnonzero(codes >= 0)
    order = keep[np.argsort(codes[keep], kind="stable")]
    counts = np.bincount(codes[keep], minlength=len(uniques))
    seg_off = np.zeros(len(uniques) + 1, dtype=np.int32)
    seg_off[1:] = np.cumsum(counts)
    values = np.ascontiguousarray(merged[feature_cols].to_numpy(dtype=np.float64))
    _lib.load()
    _lib.require_gpu()
    rows = torch.from_numpy(values).cuda()
    out = segment_mean_std(rows, order.astype(np.int32), seg_off)
    torch.cuda.synchronize()
    host = out.cpu().numpy().reshape(len(uniques), 2 * len(feature_cols))
    names = []
    for c in feature_cols:
        names += ["_".join((str(c), "mean")).strip(), "_".join((str(c), "std")).strip()]                  # :57
    final = pd.DataFrame(host, columns=names)
    final.insert(0, "unique_participant_id", list(uniques))
    return final


def aggregate_interview_sequences(clip_sequences, interview_metadata_df):
    """Drop-in for ``src/utils.py:64-104``: participant id -> the participant's clip sequences stacked in
    metadata order; clips without a sequence are skipped, participants without any are absent."""
    import torch
    groups = interview_metadata_df.groupby("unique_participant_id")["filename"].apply(list)             # :83
    print("\nAggregating interview clips into single sequences per participant...")
    names = [f for f in clip_sequences]
    if not names:
        return {}
    width = int(np.asarray(clip_sequences[names[0]]).shape[1])
    start, total = {}, 0
    for f in names:
        start[f] = total
        total += int(np.asarray(clip_sequences[f]).shape[0])
    flat = np.concatenate([np.asarray(clip_sequences[f], dtype=np.float32).reshape(-1, width) for f in names], axis=0)
    src_row, spans = [], {}
    for pid, files in groups.items():
        a = len(src_row)
        for f in files:
            if f in clip_sequences:                                                                       # :92
                n = int(np.asarray(clip_sequences[f]).shape[0])
                src_row.extend(range(start[f], start[f] + n))
        if len(src_row) > a:                                                                              # :95
            spans[pid] = (a, len(src_row))
    if not spans:
        return {}
    _lib.load()
    _lib.require_gpu()
    src = torch.from_numpy(np.ascontiguousarray(flat)).cuda()
    out = gather_rows(src, np.asarray(src_row, dtype=np.int64))
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    return {pid: host[a:b].copy() for pid, (a, b) in spans.items()}


def pad_batch_device(seq, frame_off, members):
    """Device collate: ``seq`` float32 [total_frames, width] (device) holds the clips' sequences back to back
    (``frame_off[i]`` .. ``frame_off[i+1]``); ``members`` is a list of lists of clip indices (one list per batch
    item, stacked in order).  Returns (batch float32 [B, T_max, width] right-zero-padded, lengths int list)."""
    import torch
    lens = [sum(int(frame_off[c + 1] - frame_off[c]) for c in m) for m in members]
    tmax = max(lens) if lens else 0
    idx = np.full((len(members), tmax), -1, dtype=np.int64)
    for b, m in enumerate(members):
        p = 0
        for c in m:
            n = int(frame_off[c + 1] - frame_off[c])
            idx[b, p:p + n] = np.arange(int(frame_off[c]), int(frame_off[c + 1]))
            p += n
    out = gather_rows(seq, idx.reshape(-1))
    return out.view(len(members), tmax, int(seq.shape[1])), lens
