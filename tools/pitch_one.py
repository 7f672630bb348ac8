"""One MSHDS pitch configuration in a short loop, for `rocprofv3 --pmc` passes (tool).
usage: python tools/pitch_one.py <ac_wide|ac_75|cc_hnr_75|cc_hnr_100|cc_pulses> [clips]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robust_speech_analysis_framework_amd import synth
from robust_speech_analysis_framework_amd.mshds import MshdsEngine
cfgs = {
    "ac_wide": dict(time_step=0.005, floor=50.0, ceiling=600.0),
    "ac_75": dict(time_step=0.005, floor=75.0, ceiling=500.0),
    "cc_hnr_75": dict(time_step=0.005, floor=75.0, ceiling=8000.0, silence_threshold=0.1, voicing_threshold=0.0, octave_cost=0.0,
                      octave_jump_cost=0.0, voiced_unvoiced_cost=0.0, periods=4.5, is_cc=True, refine_depth=700),
    "cc_hnr_100": dict(time_step=0.005, floor=100.0, ceiling=8000.0, silence_threshold=0.1, voicing_threshold=0.0, octave_cost=0.0,
                       octave_jump_cost=0.0, voiced_unvoiced_cost=0.0, periods=4.5, is_cc=True, refine_depth=700),
    "cc_pulses": dict(time_step=0.005, floor=75.0, ceiling=500.0, periods=1.0, is_cc=True),
}
name = sys.argv[1]
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 32
wav = torch.from_numpy(synth.synth_batch(clips, 30.0, pool=8)).cuda().reshape(-1)
n = wav.numel() // clips
offs = np.arange(clips, dtype=np.int64) * n
eng = MshdsEngine()
gp = eng.clip_peaks(wav, offs, [n] * clips)
for _ in range(3):
    eng.pitch(wav, offs, [n] * clips, gp, **cfgs[name])
torch.cuda.synchronize()
print("done", name)
