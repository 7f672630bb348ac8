import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from robust_speech_analysis_framework_amd import smile, synth
for fs, n in ((8000, 400), (16000, 400), (22050, 400), (44100, 400), (48000, 400)):
    base = np.stack([synth.synth_clip(k, 30.0, fs=fs) for k in range(4)])
    wav = torch.from_numpy(base[np.arange(n) % 4]).cuda()
    p = smile.pack_clips(wav, fs=fs)
    f = smile.smile_features(p); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2): f = smile.smile_features(p)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    print(f"fs {fs}: {n} x 30 s clips: {ms:.1f} ms per pass = {n * 30.0 / ms * 1e3:.0f} audio-s/s (whole openSMILE chain)", flush=True)
