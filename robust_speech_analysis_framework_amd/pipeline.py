"""Batch runner of the hot path: extract -> CNN-LSTM forward over one shard of clips.

This is the build's own runner for equal-length batches that are already resident in HBM
(bench.py, the multi-GPU driver).  Per clip it emits one fixed-width float32 row
``[smile 912 | ...]`` that ranks all-gather once per step.
"""
from __future__ import annotations

from . import _lib, smile

BUILT_STAGES = ["smile"]

# algorithmic traffic per audio-second of the HBM-bound kernels:
#   16 000 float32 samples read + 38 float32 LLDs x 100 frames/s written
SMILE_LLD_BYTES_PER_AUDIO_S = 16000 * 4 + 38 * 4 * 100


def resolve_stages(spec: str):
    if spec in ("all", ""):
        return list(BUILT_STAGES)
    st = [s.strip() for s in spec.split(",") if s.strip()]
    for s in st:
        if s not in BUILT_STAGES:
            raise ValueError(f"stage '{s}' is not built (built: {BUILT_STAGES})")
    return st


class Pipeline:
    def __init__(self, stages, device, seconds: float):
        _lib.load()
        self.stages = list(stages)
        self.device = device
        self.seconds = seconds
        self._packed = None
        self._packed_key = None
        self.finite_cols = None

    def _pack(self, wav):
        key = (wav.data_ptr(), tuple(wav.shape))
        if self._packed_key != key:
            self._packed = smile.pack_clips(wav, device=self.device)
            self._packed_key = key
        return self._packed

    def run(self, wav):
        """wav: float32 [clips, samples] on the device -> rows float32 [clips, width]."""
        import torch
        cols = []
        p = self._pack(wav)
        if "smile" in self.stages:
            cols.append(smile.smile_features(p))
        rows = cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)
        if self.finite_cols is None:
            self.finite_cols = torch.isfinite(rows[0]).nonzero().flatten()
        return rows

    def describe(self, clips, seconds):
        parts = []
        if "smile" in self.stages:
            parts.append("openSMILE-style 32/38 LLD + 912 functionals")
        return (f"{' + '.join(parts)} on {clips} x {seconds:g} s synthetic 16 kHz mono clips per GPU "
                f"(BASELINE config 2 shape; stages not built yet are listed in DESIGN.md)")


def roofline(prof, pipe, clips, seconds, steps, hbm_peak_gbs, mfma_peak_tflops):
    """roofline object for the kernel family with the largest summed event time."""
    if not prof:
        return None
    name = max(prof, key=lambda k: prof[k]["ms"])
    rec = prof[name]
    avg_ms = rec["ms"] / max(rec["launches"], 1)
    if rec["flops"] > 0:
        achieved = rec["flops"] / (rec["ms"] * 1e-3) / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(achieved, 3), "peak": mfma_peak_tflops,
                "unit": "TFLOP/s", "frac": round(achieved / mfma_peak_tflops, 4), "traffic": None,
                "avg_launch_ms": round(avg_ms, 4), "launches": rec["launches"]}
    if name == "smile_lld":
        per_launch = SMILE_LLD_BYTES_PER_AUDIO_S * clips * seconds * steps / max(rec["launches"], 1)
    elif rec["bytes"] > 0:
        per_launch = rec["bytes"] / max(rec["launches"], 1)
    else:
        return {"kernel": name, "bound": "hbm", "achieved": None, "peak": hbm_peak_gbs, "unit": "GB/s",
                "frac": None, "traffic": None, "avg_launch_ms": round(avg_ms, 4)}
    achieved = per_launch / (avg_ms * 1e-3) / 1e9
    return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": hbm_peak_gbs,
            "unit": "GB/s", "frac": round(achieved / hbm_peak_gbs, 5), "traffic": None,
            "algorithmic_bytes_per_launch": per_launch, "avg_launch_ms": round(avg_ms, 4),
            "launches": rec["launches"]}
