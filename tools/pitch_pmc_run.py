"""One pass of the four pitch parameter sets the extractor uses on 64 x 30 s clips (workload for the PMC passes of the MSHDS pitch kernels;
tools/pitch_phase.py times the phases of the same calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robust_speech_analysis_framework_amd import synth
from robust_speech_analysis_framework_amd.mshds import MshdsEngine

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
wav = torch.from_numpy(synth.synth_batch(clips, 30.0, pool=8)).to(dev)
n_s = wav.shape[1]
offs = np.arange(clips, dtype=np.int64) * n_s
lens = [n_s] * clips
eng = MshdsEngine(dev)
flat = wav.reshape(-1)
gpeak = eng.clip_peaks(flat, offs, lens)
hnr = dict(time_step=0.005, ceiling=8000.0, max_candidates=15, silence_threshold=0.1, voicing_threshold=0.0, octave_cost=0.0,
           octave_jump_cost=0.0, voiced_unvoiced_cost=0.0, periods=4.5, is_cc=True, refine_depth=700)
for kw in (dict(time_step=0.005, floor=50.0, ceiling=600.0), dict(time_step=0.005, floor=100.0, ceiling=500.0, voicing_threshold2=0.3),
           dict(floor=60.0, **hnr), dict(floor=100.0, **hnr), dict(time_step=0.005, floor=75.0, ceiling=500.0, periods=1.0, is_cc=True, refine_depth=70)):
    for _ in range(2):
        eng.pitch(flat, offs, lens, gpeak, **kw)
torch.cuda.synchronize()
print("done")
