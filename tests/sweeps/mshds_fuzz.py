"""One-off randomized parity sweep of the MSHDS HIP path against the CPU oracle (tool; tests/ holds the fixed cases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import mshds_oracle as mo
from robust_speech_analysis_framework_amd import synth
from robust_speech_analysis_framework_amd.mshds import MshdsEngine, FEATURE_NAMES

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.Generator(np.random.PCG64(first))
clips = [synth.synth_clip(first + k, float(rng.uniform(float(os.environ.get("FUZZ_MIN_S", "1.2")), float(os.environ.get("FUZZ_MAX_S", "3.5"))))) for k in range(count)]
lens = [len(c) for c in clips]
offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
wav = torch.from_numpy(np.concatenate(clips)).cuda()
eng = MshdsEngine()
out, ranges = eng.extract_packed(wav, [int(o) for o in offs], lens)
torch.cuda.synchronize()
got = out.cpu().numpy()
worst = 0.0
bad = 0
t0 = time.time()
for i, c in enumerate(clips):
    ref, rng_ref = mo.extract(c)
    ok_rng = tuple(rng_ref) == tuple(ranges[i])
    nanok = np.array_equal(np.isnan(got[i]), np.isnan(ref))
    m = ~np.isnan(ref) & ~np.isnan(got[i])
    rel = np.abs(got[i][m] - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3)
    w = rel.max() if m.any() else 0.0
    worst = max(worst, w)
    flag = "" if (ok_rng and nanok and w <= 1e-4) else "  <-- MISMATCH"
    if flag:
        bad += 1
        j = int(np.argmax(rel)) if m.any() else -1
        names = [n for n, mm in zip(FEATURE_NAMES, m) if mm]
        print(f"clip {first + i}: range {ranges[i]} vs {rng_ref}, nan pattern {nanok}, worst rel {w:.3e} at {names[j] if j >= 0 else '-'}{flag}", flush=True)
    else:
        print(f"clip {first + i}: ok (range {ranges[i]}, worst rel {w:.2e}) [{time.time() - t0:.0f} s]", flush=True)
print(f"SUMMARY clips {count} mismatches {bad} worst_rel {worst:.3e}")
