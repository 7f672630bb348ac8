"""Batch runner of the hot path: extract -> CNN-LSTM forward over one shard of clips.

This is the build's own runner for equal-length batches that are already resident in HBM
(bench.py, the multi-GPU driver).  Per clip it emits one fixed-width float32 row
``[mshds 25 | smile 912 | logits 2 | w2v2 frames 1]`` that ranks all-gather once per step.
The Wav2Vec2 sequences (5.66 MB per 30 s clip) stay on the producing GPU and feed its local
CNN-LSTM, exactly as notebook 03 feeds the classifier with the extractor's output.
"""
from __future__ import annotations

import numpy as np

from . import _lib, smile

BUILT_STAGES = ["mshds", "smile", "w2v2", "cnnlstm"]     # bench config C4 ("cnnlstm_only") runs the classifier without a Pipeline

# algorithmic traffic per audio-second of the HBM-bound kernels:
#   16 000 float32 samples read + 38 float64 LLDs x 100 frames/s written (32 by the frame kernel; the chain is float64)
SMILE_LLD_BYTES_PER_AUDIO_S = 16000 * 4 + 38 * 8 * 100
LDS_PEAK_GBS = 256 * 128 * 2.4        # 256 CUs x 128 B per clock x 2.4 GHz = 78.6 TB/s aggregate LDS bandwidth (MI355X_MICROARCH.md)
LATENCY_FAMILIES = {                  # one dependent step per frame: the figure that matters is the time per step
    "mshds_pitch_path": "Viterbi over <= 15 candidates, one wave per clip",
    "smile_viterbi": "Viterbi over 7 states, one wave per clip",
    "lstm_recurrent": "persistent bi-LSTM recurrence",
}
F16_MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak
SPLIT_PRODUCTS = 3                  # gemm_f16x3: fp16 MFMA products executed per algorithmic multiply-add


def resolve_stages(spec: str):
    if spec in ("all", ""):
        return list(BUILT_STAGES)
    st = [s.strip() for s in spec.split(",") if s.strip()]
    for s in st:
        if s not in BUILT_STAGES:
            raise ValueError(f"stage '{s}' is not built (built: {BUILT_STAGES})")
    if "cnnlstm" in st and "w2v2" not in st:
        raise ValueError("stage 'cnnlstm' consumes the output of stage 'w2v2'")
    return st


class Pipeline:
    def __init__(self, stages, device, seconds: float, w2v2_seed: int = 0, cnnlstm_seed: int = 0,
                 w2v2_chunks_per_call: int = 256, overlap: bool = True):
        import torch
        _lib.load()
        self.stages = list(stages)
        self.device = device
        self.seconds = seconds
        self._packed = None
        self._packed_key = None
        self.finite_cols = None
        self.last_frames = None            # Wav2Vec2 frames per clip of the last run (CNN-LSTM sequence length)
        self.w2v2 = None
        self.model = None
        self.mshds = None
        # MSHDS is fp64-VALU work, Wav2Vec2/CNN-LSTM is MFMA work: different pipes of the same CUs.  With
        # `overlap` the two run on separate HIP streams so the dispatcher co-schedules their workgroups.
        self.overlap = overlap and "mshds" in self.stages and "w2v2" in self.stages
        import os
        prio = int(os.environ.get("RSAF_AUX_PRIORITY", "0"))      # experiment knob: -1 = high priority for the MSHDS stream
        self._aux = torch.cuda.Stream(device=device, priority=prio) if self.overlap else None
        self._pool = None
        if self.overlap:
            # the main stream's launch queue back-pressures its host thread (thousands of launches per
            # step), so the auxiliary stream is driven by its own host thread
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rsaf-mshds")
        if "mshds" in self.stages:
            from .mshds import MshdsEngine
            self.mshds = MshdsEngine(device)
        if "w2v2" in self.stages:
            from .w2v2 import W2V2Engine
            from .w2v2_config import W2V2Config, random_state_dict
            cfg = W2V2Config()
            self.w2v2 = W2V2Engine(cfg, random_state_dict(cfg, w2v2_seed), device,
                                   max_chunks_per_call=w2v2_chunks_per_call)
        if "cnnlstm" in self.stages:
            from .cnnlstm import CNNLSTM
            torch.manual_seed(cnnlstm_seed)
            self.model = CNNLSTM().to(device).eval()          # reference defaults: C = H = 128, silu

    def _pack(self, wav):
        key = (wav.data_ptr(), tuple(wav.shape))
        if self._packed_key != key:
            self._packed = smile.pack_clips(wav, device=self.device)
            self._packed_key = key
        return self._packed

    def run(self, wav, only=None):
        """wav: float32 [clips, samples] on the device -> rows float32 [clips, width].  ``only``: run this subset of the
        pipeline's stages (bench.py's per-config lines reuse one pipeline: C2 = mshds + smile, C3 = w2v2)."""
        import torch
        if only is not None:
            keep = (self.stages, self.mshds, self.w2v2, self.model, self.overlap)
            self.stages = [s for s in self.stages if s in only]
            self.mshds = self.mshds if "mshds" in only else None
            self.w2v2 = self.w2v2 if "w2v2" in only else None
            self.model = self.model if "cnnlstm" in only else None
            self.overlap = False
            try:
                return self.run(wav)
            finally:
                self.stages, self.mshds, self.w2v2, self.model, self.overlap = keep
        cols = []
        p = self._pack(wav)
        n_clips, n_samp = int(wav.shape[0]), int(wav.shape[1])
        main = torch.cuda.current_stream()
        mshds_cols = None
        smile_cols = None
        if self.mshds is not None and not self.overlap:
            offs = np.arange(n_clips, dtype=np.int64) * n_samp
            # the openSMILE-style stage does not depend on MSHDS: queued at MSHDS' one host wait (the speaker ranges come back to
            # the host there), it keeps the GPU busy while the host decides the ranges and queues the remaining analyses
            hold = {}

            def _smile_now():
                if "smile" in self.stages:
                    hold["cols"] = smile.smile_features(p).to(torch.float32)
            feats, _ = self.mshds.extract_packed(p.wav, offs, [n_samp] * n_clips, before_host_sync=_smile_now)
            mshds_cols = feats.to(torch.float32)
            smile_cols = hold.get("cols")
        fut = None
        if self.overlap:
            self._aux.wait_stream(main)

            def _mshds_job():
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self._aux):
                    offs_ = np.arange(n_clips, dtype=np.int64) * n_samp
                    feats_, _ = self.mshds.extract_packed(p.wav, offs_, [n_samp] * n_clips)
                    return feats_.to(torch.float32)
            fut = self._pool.submit(_mshds_job)
        if self.mshds is not None and not self.overlap:
            cols.append(mshds_cols)
        if "smile" in self.stages:
            cols.append(smile_cols if smile_cols is not None else smile.smile_features(p).to(torch.float32))
        if self.w2v2 is not None:
            offs = np.arange(n_clips, dtype=np.int64) * n_samp
            seq, frame_off = self.w2v2.extract_packed(p.wav, offs, [n_samp] * n_clips)
            frames = int(frame_off[1]) if n_clips else 0
            self.last_frames = frames
            if self.model is not None:
                logits = self.model(seq.view(n_clips, frames, self.w2v2.cfg.hidden_size))
                cols.append(logits)
            cols.append(torch.full((n_clips, 1), float(frames), dtype=torch.float32, device=self.device))
        if self.overlap:
            mshds_cols = fut.result()
            main.wait_stream(self._aux)
            mshds_cols.record_stream(main)
            cols.insert(0, mshds_cols)
        rows = cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)
        if self.finite_cols is None:
            self.finite_cols = torch.isfinite(rows[0]).nonzero().flatten()
        return rows

    def describe(self, clips, seconds):
        parts = []
        if "mshds" in self.stages:
            parts.append("MSHDS Praat-style 25/25 features (speech rate, pitch, intensity, HNR, LTAS slope/tilt, CPPS, formants, spectral moments; fp64)")
        if "smile" in self.stages:
            parts.append("openSMILE-style 38/38 LLD (incl. SHS pitch + Viterbi, jitter / shimmer / logHNR) + 912 functionals")
        if "w2v2" in self.stages:
            parts.append("Wav2Vec2-base frame embeddings (5 s windows / 4 s hop, fp32, seeded random weights)")
        if "cnnlstm" in self.stages:
            parts.append("CNN-LSTM-attn forward (C=H=128) on the Wav2Vec2 sequences")
        return f"{' -> '.join(parts)} on {clips} x {seconds:g} s synthetic 16 kHz mono clips per GPU"


def rooflines(prof, stages, clips, seconds, steps, hbm_peak_gbs, mfma_f32_peak_tflops, f64_peak_tflops, wall_ms=None,
              frames_per_launch=None):
    """roofline objects of the profiled kernel families that have an algorithmic work model, largest event time first
    (the first one is the line's ``roofline``).  achieved = algorithmic FLOPs (or bytes) per launch / average launch
    time from HIP events on the launch stream.  ``share_of_step_wall`` = the family's event time / the timed region's
    wall time; the MSHDS analyses run on two HIP streams beside each other, so the shares of concurrent families add up
    to more than 1 (their event times overlap).  ``frames_per_launch``: {family: dependent steps per launch} for the
    latency-bound families, which get ``us_per_step``."""
    out = []
    for name, rec in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        if rec["launches"] <= 0 or rec["ms"] <= 0:
            continue
        avg_ms = rec["ms"] / rec["launches"]
        base = {"kernel": name, "avg_launch_ms": round(avg_ms, 4), "launches": rec["launches"], "traffic": None,
                "share_of_step_wall": round(rec["ms"] / wall_ms, 4) if wall_ms else None}
        if name in LATENCY_FAMILIES and name != "lstm_recurrent":
            r = {**base, "bound": "latency", "what": LATENCY_FAMILIES[name]}
            if frames_per_launch and frames_per_launch.get(name):
                r["steps_per_launch"] = int(frames_per_launch[name])
                r["us_per_step"] = round(1e3 * avg_ms / frames_per_launch[name], 3)
            out.append(r)
            continue
        if name in ("mshds_pitch_ac_fft", "mshds_pitch_cc_fft"):
            # One wave per frame (csrc/wave_fft.h): the transforms run in registers and make two trips through LDS each, so
            # the kernel's ceiling is the fp64 vector rate; the LDS traffic it still has is kept beside it.
            per_launch = rec["bytes"] / rec["launches"]
            lds = per_launch / (avg_ms * 1e-3) / 1e9
            tf = rec["flops"] / (rec["ms"] * 1e-3) / 1e12
            out.append({**base, "bound": "fp64_vector", "achieved": round(tf, 3), "peak": f64_peak_tflops, "unit": "TFLOP/s",
                        "frac": round(tf / f64_peak_tflops, 4), "algorithmic_flops_per_launch": rec["flops"] / rec["launches"],
                        "lds_gbs": round(lds, 1), "lds_frac_of_peak": round(lds / LDS_PEAK_GBS, 4),
                        "algorithmic_lds_bytes_per_launch": per_launch,
                        "note": "FLOPs = 5 S log2 S per complex transform + ~30 per point of the spectrum step (real operations, an "
                                "FMA counts two); LDS bytes = the two exchanges of each transform (16 B x S written and read) + the "
                                "paired spectrum step; phase timing: profiles/r03/pitch_phase_r03.txt"})
            continue
        if rec["flops"] > 0:
            fp64 = name.startswith("mshds_")
            split3 = name in ("w2v2_gemm", "w2v2_posconv_gemm")      # gemm_f16x3 / posconv_f16x3_kernel: the same three-product arithmetic
            peak = f64_peak_tflops if fp64 else (F16_MFMA_PEAK_TFLOPS if split3 else mfma_f32_peak_tflops)
            alg = rec["flops"] / (rec["ms"] * 1e-3) / 1e12
            # the f16x3 GEMM executes three fp16 MFMA products per algorithmic multiply-add (two-way fp16 splits of both
            # operands): its roofline is the fp16 matrix pipe, priced with the FLOPs the pipe actually executes; the
            # algorithmic (fp32-equivalent) rate is kept beside it
            ach = SPLIT_PRODUCTS * alg if split3 else alg
            r = {**base, "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                 "arithmetic": ("f64 (v_mfma_f64_16x16x4_f64 issues at the fp64 vector rate)" if fp64 else
                                ("fp32-accurate result from 3 fp16 MFMA products of two-way operand splits (power-of-two row "
                                 "scales), fp32 accumulation; rounds 2-3 needed 6 bf16 products for the same result"
                                 if split3 else "f32 MFMA")),
                 "algorithmic_flops_per_launch": rec["flops"] / rec["launches"]}
            if split3:
                r["fp32_equivalent_tflops"] = round(alg, 3)
                r["fp32_equivalent_over_fp32_mfma_peak"] = round(alg / mfma_f32_peak_tflops, 4)
                r["executed_products_per_multiply_add"] = SPLIT_PRODUCTS
            if name == "lstm_recurrent":
                r["note"] = "latency-bound persistent recurrence: the figure that matters is the time per step"
            out.append(r)
        elif name == "smile_lld":
            per_launch = SMILE_LLD_BYTES_PER_AUDIO_S * clips * seconds * steps / rec["launches"]
            ach = per_launch / (avg_ms * 1e-3) / 1e9
            out.append({**base, "bound": "hbm", "achieved": round(ach, 2), "peak": hbm_peak_gbs, "unit": "GB/s",
                        "frac": round(ach / hbm_peak_gbs, 5), "algorithmic_bytes_per_launch": per_launch,
                        "note": "SURVEY.md 8d assigns HBM (64 000 B read + 30 400 B of float64 LLD rows written per audio-second); the "
                                "kernel is float64 since round 3 (every decision of the chain coincides with the oracle) and is "
                                "bound by fp64 issue + LDS round trips at 2 waves / SIMD: profiles/r03/smile_phase.txt, DESIGN.md"})
        elif rec["bytes"] > 0:
            per_launch = rec["bytes"] / rec["launches"]
            ach = per_launch / (avg_ms * 1e-3) / 1e9
            out.append({**base, "bound": "hbm", "achieved": round(ach, 2), "peak": hbm_peak_gbs, "unit": "GB/s",
                        "frac": round(ach / hbm_peak_gbs, 5), "algorithmic_bytes_per_launch": per_launch})
    return out


def c2_level_roofline(prof_one, clips, seconds, wall_s, hbm_peak_gbs, f64_peak_tflops):
    """One entry for BASELINE config C2 as a whole (MSHDS + openSMILE-style stages, one-stream pass).  SURVEY.md 8d assigns
    the stage the HBM roof at 64 000 B per audio-second (the waveform read once): that figure is kept, and beside it the
    fp64 work model the stage really lives on - the summed algorithmic FLOPs of its families that carry one (correlation
    FFTs, Chebyshev coefficient builds: Praat's arithmetic is float64 on the vector / fp64 matrix pipe) over the pass."""
    audio_s = clips * seconds
    flops = sum(v["flops"] for k, v in prof_one.items() if k.startswith("mshds") and v["flops"] > 0)
    counted = sorted(k for k, v in prof_one.items() if k.startswith("mshds") and v["flops"] > 0)
    tf = flops / wall_s / 1e12
    hbm = 64000.0 * audio_s / wall_s / 1e9
    return {"kernel": "C2 (MSHDS + openSMILE-style) as a whole, one-stream pass", "wall_ms": round(1e3 * wall_s, 3),
            "bound": "fp64_vector", "achieved": round(tf, 3), "peak": f64_peak_tflops, "unit": "TFLOP/s",
            "frac": round(tf / f64_peak_tflops, 4), "algorithmic_fp64_flops": flops, "families_counted": counted,
            "families_not_counted": "Brent iterations, path finder, pulse walker, Burg / root finder, CPPS, intensity, "
                                    "openSMILE-style chain (latency- or issue-bound, no closed FLOP count)",
            "survey_8d_hbm_reading": {"bytes_per_audio_s": 64000, "achieved_GBs": round(hbm, 2), "peak_GBs": hbm_peak_gbs,
                                      "frac": round(hbm / hbm_peak_gbs, 6),
                                      "note": "the stage reads each sample a few dozen times from L2 / LDS and computes in fp64: "
                                              "the HBM roof SURVEY.md assigned is not what bounds it"}}
