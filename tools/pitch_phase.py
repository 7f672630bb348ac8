"""Phase timing of the MSHDS pitch frame kernel (profiling aid; not part of the product path).

Runs rsaf_mshds_pitch with env RSAF_PITCH_STOP=k (leave the frame kernel after phase k) for the four
parameter sets the extractor uses and prints the frame-kernel event time per (config, k).
  1 = segment load + local mean/peak   2 = + correlation   3 = + normalise + maxima
  4 = + first estimates (sinc 30)      5 = + candidate list   6 = + Chebyshev coefficients of the candidates' cells
  0 = everything (+ Brent refinement)
"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robust_speech_analysis_framework_amd import _lib, synth
from robust_speech_analysis_framework_amd.mshds import MshdsEngine

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
stops = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2, 3, 4, 5, 6, 0)
dev = torch.device("cuda:0")
wav = torch.from_numpy(synth.synth_batch(clips, 30.0, pool=8)).to(dev)
n_s = wav.shape[1]
offs = np.arange(clips, dtype=np.int64) * n_s
lens = [n_s] * clips
eng = MshdsEngine(dev)
flat = wav.reshape(-1)
gpeak = eng.clip_peaks(flat, offs, lens)
cfgs = {
    "ac_wide_50_600": dict(time_step=0.005, floor=50.0, ceiling=600.0),
    "ac_75_500": dict(time_step=0.005, floor=75.0, ceiling=500.0),
    "cc_hnr_75": dict(time_step=0.005, floor=75.0, ceiling=8000.0, max_candidates=15, silence_threshold=0.1,
                      voicing_threshold=0.0, octave_cost=0.0, octave_jump_cost=0.0, voiced_unvoiced_cost=0.0,
                      periods=4.5, is_cc=True, refine_depth=700),
    "cc_pulses_75_500": dict(time_step=0.005, floor=75.0, ceiling=500.0, periods=1.0, is_cc=True, refine_depth=70),
    "cc_hnr_60": dict(time_step=0.005, floor=60.0, ceiling=8000.0, max_candidates=15, silence_threshold=0.1,
                      voicing_threshold=0.0, octave_cost=0.0, octave_jump_cost=0.0, voiced_unvoiced_cost=0.0,
                      periods=4.5, is_cc=True, refine_depth=700),
    "cc_hnr_100": dict(time_step=0.005, floor=100.0, ceiling=8000.0, max_candidates=15, silence_threshold=0.1,
                       voicing_threshold=0.0, octave_cost=0.0, octave_jump_cost=0.0, voiced_unvoiced_cost=0.0,
                       periods=4.5, is_cc=True, refine_depth=700),
    "ac_speechrate_30_450": dict(time_step=0.02, floor=30.0, ceiling=450.0, max_candidates=4, silence_threshold=0.03, voicing_threshold=0.25,
                                 octave_cost=0.01, octave_jump_cost=0.35, voiced_unvoiced_cost=0.25),
    "ac_60_250_dual": dict(time_step=0.005, floor=60.0, ceiling=250.0, voicing_threshold2=0.3),
    "ac_100_500_dual": dict(time_step=0.005, floor=100.0, ceiling=500.0, voicing_threshold2=0.3),
}
res = {}
for name, kw in cfgs.items():
    if only and name not in only:
        continue
    for stop in stops:
        os.environ["RSAF_PITCH_STOP"] = str(stop)
        eng.pitch(flat, offs, lens, gpeak, **kw)           # warm
        torch.cuda.synchronize()
        _lib.prof_begin()
        eng.pitch(flat, offs, lens, gpeak, **kw)
        torch.cuda.synchronize()
        pr = _lib.prof_end()
        ms = sum(v["ms"] for k, v in pr.items() if k.startswith("mshds_pitch_") and k != "mshds_pitch_path")
        res[f"{name}/stop{stop}"] = round(ms, 3)
        print(name, "stop", stop, "frame kernel ms", round(ms, 3), "path ms", round(pr.get("mshds_pitch_path", {}).get("ms", 0), 3), flush=True)
        if stop == 0:
            print("   ", {k[len("mshds_pitch_"):]: round(v["ms"], 3) for k, v in pr.items() if k.startswith("mshds_pitch_")}, flush=True)
os.environ.pop("RSAF_PITCH_STOP", None)
print(json.dumps(res))
