"""Edge-input sweep of the MSHDS HIP path against the oracle: degenerate signals must give the same NaN pattern / values
and must never fault.  ``run()`` is called by tests/test_sweeps_gpu.py; ``python tests/sweeps/mshds_edge.py`` prints the table."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def run():
    import torch
    from oracle import mshds_oracle as mo
    from robust_speech_analysis_framework_amd import synth
    from robust_speech_analysis_framework_amd.mshds import MshdsEngine, FEATURE_NAMES
    rng = np.random.Generator(np.random.PCG64(77))
    fs = 16000
    t = np.arange(2 * fs) / fs
    cases = {
        "zeros_2s": np.zeros(2 * fs),
        "dc_0.3_2s": np.full(2 * fs, 0.3),
        "white_noise_2s": 0.1 * rng.standard_normal(2 * fs),
        "sine_100Hz_2s": 0.4 * np.sin(2 * np.pi * 100 * t),
        "sine_300Hz_2s": 0.4 * np.sin(2 * np.pi * 300 * t),
        "sine_40Hz_2s": 0.4 * np.sin(2 * np.pi * 40 * t),
        "impulse_train_125Hz_2s": (np.arange(2 * fs) % 128 == 0) * 0.8,
        "square_200Hz_fullscale_1s": np.sign(np.sin(2 * np.pi * 200 * t[:fs])) * 0.999,
        "tiny_amplitude_speech_2s": synth.synth_clip(400, 2.0).astype(np.float64) * 1e-5,
        "speech_400_samples": synth.synth_clip(401, 0.025),
        "speech_1000_samples": synth.synth_clip(402, 0.0625),
        "speech_3000_samples": synth.synth_clip(403, 0.1875),
        "speech_8000_samples": synth.synth_clip(404, 0.5),
        "speech_then_silence": np.concatenate([synth.synth_clip(405, 1.0), np.zeros(fs)]),
        "silence_then_speech": np.concatenate([np.zeros(fs), synth.synth_clip(406, 1.0)]),
        "one_sample": np.array([0.5]),
        "chirp_80_400Hz_2s": 0.4 * np.sin(2 * np.pi * (80 * t + 80 * t * t)),
    }
    eng = MshdsEngine()
    bad = 0
    for name, x in cases.items():
        x = np.ascontiguousarray(x, dtype=np.float32)
        try:
            wav = torch.from_numpy(x).cuda()
            out, ranges = eng.extract_packed(wav, [0], [len(x)])
            torch.cuda.synchronize()
            got = out.cpu().numpy()[0]
        except Exception as e:
            print(f"{name}: GPU path raised {type(e).__name__}: {e}")
            got = None
        try:
            ref, rr = mo.extract(x)
        except Exception as e:
            print(f"{name}: oracle raised {type(e).__name__}: {e}")
            ref = None
        if got is None or ref is None:
            status = "both raised" if (got is None and ref is None) else "ONLY ONE SIDE RAISED  <-- MISMATCH"
            bad += status.endswith("MISMATCH")
            print(f"{name}: {status}", flush=True)
            continue
        nanok = np.array_equal(np.isnan(got), np.isnan(ref))
        m = ~np.isnan(ref) & ~np.isnan(got)
        rel = np.abs(got[m] - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3) if m.any() else np.zeros(1)
        ok = nanok and tuple(rr) == tuple(ranges[0]) and rel.max() <= 1e-4
        bad += (not ok)
        extra = ""
        if not ok:
            names = [n for n, mm in zip(FEATURE_NAMES, m) if mm]
            j = int(np.argmax(rel))
            diffnan = [n for n, a, b in zip(FEATURE_NAMES, np.isnan(got), np.isnan(ref)) if a != b]
            extra = f"  <-- MISMATCH ranges {ranges[0]} vs {rr}; nan diff {diffnan}; worst {names[j] if names else '-'} got {got[m][j] if names else ''} ref {ref[m][j] if names else ''}"
        print(f"{name}: n={len(x)} finite {int(m.sum())}/25 worst rel {rel.max():.2e}{extra}", flush=True)
    print(f"SUMMARY cases {len(cases)} mismatches {bad}")
    return {"cases": len(cases), "mismatches": int(bad)}


if __name__ == "__main__":
    sys.exit(1 if run()["mismatches"] else 0)
